#!/usr/bin/env python3
"""BASELINE configs[3] through the reference's own sequence: ContextBuilder<CKKS>().m(65536)
.precision(20).bits(1400).c(3) (benchmarks/ckks_common.h:45-50 shape), a key pair with its
relinearisation matrix, B independent pairs of PubKey::CKKSencrypt ciphertexts packed along the
batch axis, then Ctxt::multiplyBy (src/Ctxt.cpp:1681-1774 with the CKKS branches, mirrored in
helib_amd/ctxt.py) timed on the device:

  level 1   fresh x fresh: no mod-switch (a fresh CKKS ciphertext has no noise to scale down),
            tensorProduct + relin_CKKS_adjust + key switch at the full level;
  level 2   product x product: both operands are mod-switched first (bringToSet), then the same.

The last product of each level is decrypted and decoded (raw / ratFactor) and compared with the real
negacyclic product of the plaintexts.  One JSON line on stdout.

  python tools/bench_levels.py                      # MI355X, m=65536 bits=1400 batch 64
(the control flow of run() is exercised on the CPU by tests/test_keys_host.py, which hands it the
test oracle's backend; this tool itself only knows the device)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def negacyclic(a, b):
    n = len(a)
    full = np.convolve(a, b)
    return full[:n] - np.append(full[n:], 0.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=65536)
    ap.add_argument("--bits", type=int, default=1400)
    ap.add_argument("--precision", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bounds", action="store_true", help="noise bounds instead of measured noise")
    ap.add_argument("--scheme", default="ckks", choices=["ckks", "bgv"],
                    help="bgv: the same two levels for ContextBuilder<BGV>().m(m).p(p).bits(bits) -- level 2 is the "
                         "multiply a computation spends its time in (special primes and more dropped per operand)")
    ap.add_argument("--p", type=int, default=65537)
    ap.add_argument("--l1-steps", type=int, default=0, help="timed level-1 steps (default: --steps)")
    ap.add_argument("--phases", action="store_true",
                    help="synchronise around the phases of every multiply and report where the wall time goes "
                         "(bring-to-set / tensor / relinearise; device-synchronous, so slower than the pipelined run)")
    args = ap.parse_args()
    ckks = args.scheme == "ckks"

    from helib_amd import ctxt as hc, keys as hk
    if ckks:
        cc = hc.ChainContext(args.m, -1, args.precision, bits=args.bits, c=3, ckks=True)
    else:
        cc = hc.ChainContext(args.m, args.p, 1, bits=args.bits, c=3)
    import torch
    from helib_amd import capi as hx
    if not torch.cuda.is_available():
        raise SystemExit("bench_levels.py needs an MI355X (no CPU path)")
    ctx = hx.Context(cc.m, 0)
    for q in cc.primes:
        ctx.add_prime(q)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    line = run(cc, hk.HxBackend(ctx, cc), torch.cuda.synchronize,
               lambda idx, rows: hx.DoubleCRT(ctx, idx, rows.shape[1], rows), args, "hx")
    print(json.dumps(line))
    if not line["verified"]:
        raise SystemExit("bench_levels: the decrypted / decoded product is off -- results are wrong")


def run(cc, be, sync, make, args, backend):
    """The two levels over backend `be` (make(idx, rows[nrows, batch, phim]) builds its DoubleCRT);
    returns the result line."""
    from helib_amd import ctxt as hc, keys as hk
    ckks = args.scheme == "ckks"
    n, B = cc.phim, args.batch
    hc.Ctxt.measure = not args.bounds
    phase_ms = {}
    if args.phases:   # (class-level wrappers: meant for the one-shot tool process)
        def wrap(owner, name, static=False):
            fn = getattr(owner, name)

            def timed_fn(*a, **k):
                sync()
                t0 = time.perf_counter()
                r = fn(*a, **k)
                t1 = time.perf_counter()
                sync()
                t2 = time.perf_counter()
                e = phase_ms.setdefault(name, [0, 0.0, 0.0])
                e[0] += 1
                e[1] += (t1 - t0) * 1e3      # host time until the call returns
                e[2] += (t2 - t0) * 1e3      # until the device is idle again
                return r
            setattr(owner, name, staticmethod(timed_fn) if static else timed_fn)
        wrap(hc.Ctxt, "_bringManyToSet", True)
        wrap(hc.Ctxt, "_bringBothToSet", True)
        wrap(hc.Ctxt, "_tensorProduct")
        wrap(hc.Ctxt, "reLinearize")
        wrap(hc.Ctxt, "clone")
    sk = hk.SecKey(cc, be, 3)
    t0 = time.perf_counter()
    sk.GenSecKey(maxDegKswitch=2)
    sync()
    t_keygen = time.perf_counter() - t0

    rng = np.random.default_rng(11)
    f = float(1 << args.precision)
    vals = rng.uniform(-1, 1, size=(2, B, n)) / n                 # |canonical embedding| <= 1
    msgs = rng.integers(0, args.p, size=(2, B, n))                # bgv plaintexts
    L = len(cc.ctxtPrimes)
    rows = np.empty((2, 2, L, B, n), dtype=np.uint64)
    first, t_enc = None, 0.0
    for j in range(2):
        for b in range(B):
            t0 = time.perf_counter()
            ct = sk.CKKSencrypt(np.rint(vals[j, b] * f).astype(np.int64), 1.0, f) if ckks else sk.Encrypt(msgs[j, b])
            sync()
            t_enc += time.perf_counter() - t0
            if first is None:
                first = ct
            assert abs(ct.lnRatFactor - first.lnRatFactor) < 1e-12 and ct.lnNoise == first.lnNoise
            rows[j, 0, :, b] = np.asarray(ct.parts["1"].download()).reshape(L, -1, n)[:, 0]
            rows[j, 1, :, b] = np.asarray(ct.parts["s"].download()).reshape(L, -1, n)[:, 0]
    ops = []
    for j in range(2):
        c = first.clone()
        c.parts = {"1": make(list(cc.ctxtPrimes), rows[j, 0]), "s": make(list(cc.ctxtPrimes), rows[j, 1])}
        ops.append(c)
    fa, fb = ops

    def decode(ct, b):
        one = ct.clone()
        one.parts = {}
        for h, q in ct.parts.items():
            d = np.asarray(q.download())
            d = d.reshape(len(q.getIndexSet()), -1, n)[:, b:b + 1]
            one.parts[h] = make(q.getIndexSet(), d)
        raw = sk.Decrypt(one)
        if not ckks:
            return np.array(raw, dtype=np.int64)
        return np.array([float(v) for v in raw]) / math.exp(ct.lnRatFactor)

    def timed(pairs_of, steps):
        pairs = pairs_of(max(1, args.warmup))
        for a, b in pairs:
            a.multiplyBy(b)
            _ = a.lnNoise
        pairs = pairs_of(steps)
        sync()
        t0 = time.perf_counter()
        prev = None
        pairs.reverse()
        while pairs:                  # results are dropped as the loop goes (the reference's loop
            a, b = pairs.pop()        # overwrites one ciphertext), so their storage is recycled
            a.multLowLvl(b, destructive=True)
            a.reLinearize()
            if prev is not None:
                _ = prev.lnNoise
            prev = a
            del a, b
        _ = prev.lnNoise
        sync()
        return time.perf_counter() - t0, prev

    s1 = args.l1_steps or args.steps
    dt1, p1 = timed(lambda k: [(fa.clone(), fb.clone()) for _ in range(k)], s1)
    phases1 = {k: [v[0], round(v[1], 2), round(v[2], 2)] for k, v in phase_ms.items()}
    phase_ms.clear()
    if not ckks:
        def modmul(a, b):
            full = np.convolve(a.astype(object), b.astype(object))
            return np.array([int(v) for v in (full[:n] - np.append(full[n:], 0))], dtype=object) % args.p
        want1 = [modmul(msgs[0, b], msgs[1, b]) for b in (0, B - 1)]
        ok1 = all(np.array_equal(decode(p1, b).astype(object), w) for b, w in zip((0, B - 1), want1))
        dt2, p2 = timed(lambda k: [(p1.clone(), p1.clone()) for _ in range(k)], args.steps)
        ok2 = all(np.array_equal(decode(p2, b).astype(object), modmul(w, w)) for b, w in zip((0, B - 1), want1))
        line = {"tool": "bench_levels", "backend": backend, "scheme": "bgv",
                "workload": f"BGV m={cc.m} p={args.p} bits={args.bits}: L={L}, K={len(cc.specialPrimes)}, "
                            f"D={len(cc.digits)}, batch {B}, noise {'bounds' if args.bounds else 'measured'}",
                "level1_fresh_mult_per_s": round(B * s1 / dt1, 1), "level1_ms_per_step": round(dt1 / s1 * 1e3, 3),
                "level2_mult_per_s": round(B * args.steps / dt2, 1),
                "level2_ms_per_step": round(dt2 / args.steps * 1e3, 3),
                "level2_operand_primes": sorted(p1.primeSet), "level2_result_primes": sorted(p2.primeSet),
                "verified": bool(ok1 and ok2)}
        if args.phases:
            line["phases_level1"] = phases1
            line["phases_level2"] = {k: [v[0], round(v[1], 2), round(v[2], 2)] for k, v in phase_ms.items()}
        return line
    # the encoded plaintexts are rint(v*f)/f: compare with THEIR product, so that what is left is
    # the scheme's error, which must stay below the bound the ciphertext itself reports
    # (noiseBound / ratFactor; a coefficient is at most the canonical-embedding norm for m = 2^k)
    enc = np.rint(vals * f) / f
    want1 = [negacyclic(enc[0, b], enc[1, b]) for b in (0, B - 1)]
    err1 = max(float(np.max(np.abs(decode(p1, b) - w))) for b, w in zip((0, B - 1), want1))
    dt2, p2 = timed(lambda k: [(p1.clone(), p1.clone()) for _ in range(k)], args.steps)
    want2 = [negacyclic(w, w) for w in want1]
    err2 = max(float(np.max(np.abs(decode(p2, b) - w))) for b, w in zip((0, B - 1), want2))
    tol = [math.exp(p1.lnNoise - p1.lnRatFactor), math.exp(p2.lnNoise - p2.lnRatFactor)]
    mag = [float(np.max(np.abs(want1[0]))), float(np.max(np.abs(want2[0])))]
    ok = err1 <= tol[0] and err2 <= tol[1] and err1 < 1e-3 * mag[0] and err2 < 1e-3 * mag[1]
    line = {
        "tool": "bench_levels", "backend": backend,
        "workload": f"CKKS m={cc.m} precision={args.precision} bits={args.bits}: L={L} ctxt primes, "
                    f"K={len(cc.specialPrimes)} special, D={len(cc.digits)}; Ctxt::multiplyBy on CKKSencrypt "
                    f"ciphertexts, batch {B}, noise {'bounds' if args.bounds else 'measured'}",
        "level1_fresh_mult_per_s": round(B * s1 / dt1, 1), "level1_ms_per_step": round(dt1 / s1 * 1e3, 3),
        "level1_primes": len(p1.primeSet),
        "level2_mult_per_s": round(B * args.steps / dt2, 1), "level2_ms_per_step": round(dt2 / args.steps * 1e3, 3),
        "level2_operand_primes_after_bringToSet": len(p2.primeSet) - len(cc.specialPrimes),
        "decode_max_abs_err": [err1, err2], "reported_error_bound": tol, "max_abs_product_coeff": mag,
        "verified": bool(ok),
        "log2_ratFactor": [round(p1.lnRatFactor / math.log(2), 2), round(p2.lnRatFactor / math.log(2), 2)],
        "log2_noise": [round(p1.lnNoise / math.log(2), 2), round(p2.lnNoise / math.log(2), 2)],
        "keygen_ms": round(t_keygen * 1e3, 2), "encrypt_ms": round(t_enc / (2 * B) * 1e3, 3),
    }
    if args.phases:   # [calls, host ms, host+device ms] summed over warm-up and timed multiplies
        line["phases_level1"] = phases1
        line["phases_level2"] = {k: [v[0], round(v[1], 2), round(v[2], 2)] for k, v in phase_ms.items()}
    return line


if __name__ == "__main__":
    main()

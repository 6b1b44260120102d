#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X.

metric   : ciphertext x ciphertext multiplications per second, including relinearisation
           (Ctxt::multiplyBy; timed region of benchmarks/bgv_basic.cpp:144-165)
host     : the timed loop is driven by the C++17 host -- include/helib_amd_ctxt.hpp / helib_amd_keys.hpp
           (HElib's Ctxt / DoubleCRT / SecKey surface over the C ABI of include/helib_amd.h) compiled
           into helib_amd/lib/libhelib_amd_host.so (helib_amd/csrc/host_session.cpp, include/helib_amd_host.h).
           Keys, encryptions, the loop `copy = ctxt1; copy.multiplyBy(ctxt2)` and the decryption of the
           results all run there; python starts the loop, synchronises, times and checks the plaintexts.
           `config.python_mirror_mult_per_s` is the same sequence driven by helib_amd/ctxt.py (secondary).
workload : (default `bgv32768`) BGV m=32768, p=65537, bits=950 -> L=16 x 60-bit ctxt primes,
           K=6 x 56-bit special primes, 6 small primes, D=3 digits (6/5/5) -- SURVEY.md Appendix B
           shape of BASELINE configs[2].  FRESH ciphertexts, the reference's own sequence:
             multLowLvl  = bringToSet x2 (mod-up by a small prime, mod-down by a ctxt prime,
                           4 parts) + tensorProduct
             reLinearize = dropSmallAndSpecialPrimes (3 parts) + addPrimesAndScale + key switch
           Inputs as the reference benchmark prepares them (benchmarks/bgv_basic.cpp:144-157): a key
           pair + relinearisation matrix and public-key encryptions of random plaintexts; after the
           timed region EVERY batch element of the last product is decrypted and compared with the
           plaintext product (config.verified).
           Added noise is MEASURED as in the reference's default build (embeddingLargestCoeff of
           the mod-switch deltas and of the key-switch digits, src/Ctxt.cpp:466-530,
           src/DoubleCRT.cpp:530-545) -- on the device, read back when the host logic next needs
           the estimate; `config.bound_noise_mult_per_s` is the same sequence with the
           reference's alternative noise bounds (no norms).
           `config.fixed_level_mult_per_s` additionally reports tensorProduct+reLinearize alone
           (hx_mul_relin, no prime-set changes), the kernel-level pipeline DESIGN.md analyses.
           `--workload ckks65536` = BASELINE configs[3]: ContextBuilder<CKKS>().m(65536).precision(1).bits(1400)
           .scale(10) (L=24, K=8, D=3; precision and scale as benchmarks/ckks_common.h:45-50; `--bits 440` = the
           reference's own big_params, benchmarks/ckks_basic.cpp:263), CKKSencrypt-ed pairs through Ctxt::multiplyBy
           (benchmarks/ckks_basic.cpp:161-180), `--global-batch 512` split over the ranks; level 2
           (product x product) reported beside it; every rank decodes and checks its slice.
step     : `--mults-per-step` (32) x [copy(ctxt1); copy.multiplyBy(ctxt2)] over the batch of
           `--batch` (128) independent ciphertext pairs resident in HBM -- the loop body of
           benchmarks/bgv_basic.cpp:158-164, which also multiplies the same two operands every
           iteration -- i.e. 4096 ciphertext multiplications per step, so that the driver's 20 steps
           time about 2.5 s of device work.  Both operand copies (the reference's copy(ctxt1), untimed
           there, and multiplyBy's own copy of `other`, src/Ctxt.cpp:1700-1745) are made inside the
           timed region, but the engine's copies are copy-on-write and the fused mod-switch reads the
           shared slab and writes elsewhere: no bytes move for them.  A result is dropped once the
           next one is complete (the reference's loop overwrites its ciphertext).
scaling  : weak (default) -- every rank multiplies its own `--batch` pairs; `--global-batch G` splits
           G pairs over the ranks instead (helib_amd.dist.shard; configs[3]: 512 pairs, 64 per GPU) =
           strong.  No data-path collective (independent ciphertexts shard across GPUs).
launch   : `bench.py --gpus N` without WORLD_SIZE in the environment spawns the N ranks itself (one
           process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set,
           RCCL for the barrier and the max-over-ranks); under torch.distributed.run it checks
           --gpus against WORLD_SIZE.  `--dry-launch` exercises exactly that launch / shard /
           barrier / aggregate path with gloo on CPUs and no engine call (tests/test_dist_cpu.py).
other legs (rank 0, skipped by --no-extras): the reference's other benchmark lines over the resident batch --
           `config.bgv_basic_ops` (benchmarks/bgv_basic.cpp:36-211) in the default line, `config.ckks_basic_ops`
           (benchmarks/ckks_basic.cpp:38-236) in the `--workload ckks65536` line, every result decrypted /
           decoded and checked --, the two levels of the other scheme, BASELINE configs[4] (Bluestein m=21845),
           one ciphertext pair at a time eagerly and as one hipGraphLaunch.
roofline : measured IN SITU: after the timed region a few more multiplies run with every kernel launch
           carrying the dispatch's own start / stop events on its stream (hx_profile_begin / _end,
           helib_amd/csrc/prof.h).
           `roofline` is the time-dominant kernel at its dominant launch shape, `achieved` = its
           algorithmic bytes per launch / its average duration inside the multiply; the whole
           per-kernel table is `config.kernels_in_situ`; the forward row transform is reported both in
           situ and back to back (`roofline.ntt_forward`).  `traffic` (HBM bytes from PMC counters)
           cannot be collected inside this process: it is the RECORDED figure of the committed
           rocprofv3 --pmc passes over the same launch shape (labelled so).
other    : --workload bgv32768_fixed (fixed-level only, synthetic rows).
cpu_baseline: the CPU oracle = a port of the reference algorithm driven through the same sequence:
           `value` on one core, as the reference benchmarks run, `all_cores` with one process per host
           core, both on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import tempfile
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0  # the same guide: practical copy ceiling (reported beside the spec fraction)

# fixed-level shapes (prime indices 0..L-1 ctxt, L..L+K-1 special)
SHAPES = {
    "bgv32768_fixed": dict(M=32768, L=16, K=6, ct_bits=60, sp_bits=56,
                           digits=[list(range(0, 6)), list(range(6, 11)), list(range(11, 16))],
                           name="BGV m=32768 N=16384 L=16x60b K=6x56b D=3 (6/5/5) "
                                "tensorProduct+reLinearize at fixed level"),
    "ckks65536": dict(M=65536, L=24, K=8, ct_bits=59, sp_bits=59,
                      digits=[list(range(0, 8)), list(range(8, 16)), list(range(16, 24))],
                      name="CKKS m=65536 N=32768 L=24x59b K=8x59b D=3 (8/8/8) "
                           "tensorProduct+reLinearize at fixed level"),
}
# module-level view of the default fixed shape (used by tools/prof_ntt.py)
M, L, K = 32768, 16, 6
DIGITS = SHAPES["bgv32768_fixed"]["digits"]


def gen_primes(shape=None):
    """PrimeGenerator(ct_bits, m) x L then PrimeGenerator(sp_bits, m) x K (product-side helpers;
    the oracle is only loaded for the cpu_baseline leg)."""
    from helib_amd import hostnt
    sh = shape or SHAPES["bgv32768_fixed"]
    g = hostnt.PrimeGen(sh["ct_bits"], sh["M"])
    primes = [g.next() for _ in range(sh["L"])]
    g2 = g if sh["sp_bits"] == sh["ct_bits"] else hostnt.PrimeGen(sh["sp_bits"], sh["M"])
    primes += [g2.next() for _ in range(sh["K"])]
    return primes


def uniform_rows(rng, primes, idx, batch, n):
    out = np.empty((len(idx), batch, n), dtype=np.uint64)
    for r, i in enumerate(idx):
        out[r] = rng.integers(0, primes[i], size=(batch, n), dtype=np.uint64)
    return out


def algorithmic_bytes_fixed(n, l, k, d):
    """compulsory traffic of tensorProduct+reLinearize at a fixed level."""
    tensor = l * 56 * n                       # 4 parts in, 3 out (scaling fused)
    ntt = d * (l + k) * 16 * n                # L inverse + D(L+K)-L forward row transforms
    ext = d * (l + k) * 8 * n                 # read each digit's own rows once, write the extension rows
    ks = (l + k) * (3 * d + 4) * 8 * n        # fused inner product incl. own-row rebuild
    return tensor + ntt + ext + ks


def algorithmic_bytes_fresh(n, l, k, d):
    """The reference sequence for fresh operands: + 4 parts x [mod-up scale 16n per row + mod-down
    (drop 1 of l+1 rows): 1 inverse + l forward transforms, delta rows 8n, (c-delta)/D 24n per
    kept row] + 3 parts x the same mod-down for dropping the small prime after the tensor product."""
    def moddown(rows_before, dropped):
        kept = rows_before - dropped
        return (dropped + kept) * 16 * n + (dropped + kept) * 8 * n + kept * 24 * n
    pre = 4 * (l * 16 * n + moddown(l + 1, 1))
    mid = 3 * moddown(l, 1)
    return pre + algorithmic_bytes_fixed(n, l, k, d) + mid


def build_native_oracle():
    from oracle import oracle as O
    so_dir = os.path.join(ROOT, "oracle")
    native = os.path.join(so_dir, "liboracle_native.so")
    if os.environ.get("HX_ORACLE_SO") == native and os.path.exists(native):
        return O      # built by the parent process (the all-cores workers must not race on the file)
    try:  # a native-tuned build of the same C file, made on the machine that runs it
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-shared", "-o",
                               native, os.path.join(so_dir, "hx_oracle.c"), "-lm"],
                              stderr=subprocess.DEVNULL)
        os.environ["HX_ORACLE_SO"] = native
        O._LIB = None
    except Exception:
        pass
    return O


def cpu_baseline_fixed(shape, primes, sample_mults):
    O = build_native_oracle()
    octx = O.Ctx(shape["M"])
    for q in primes:
        octx.add_prime(q)
    l, k = shape["L"], shape["K"]
    own, sp = list(range(l)), list(range(l, l + k))
    allp = own + sp
    rng = np.random.default_rng(99)
    n = octx.N
    kb = np.stack([uniform_rows(rng, primes, allp, 1, n)[:, 0] for _ in shape["digits"]])
    ka = np.stack([uniform_rows(rng, primes, allp, 1, n)[:, 0] for _ in shape["digits"]])
    ops = [uniform_rows(rng, primes, own, 1, n)[:, 0] for _ in range(4)]
    octx.mul_relin(own, sp, shape["digits"], *ops, kb, ka)  # warm
    t0 = time.perf_counter()
    for _ in range(sample_mults):
        octx.mul_relin(own, sp, shape["digits"], *ops, kb, ka)
    dt = time.perf_counter() - t0
    return {"value": sample_mults / dt, "unit": "mult/s", "cores": 1, "kind": "port",
            "sample": f"{sample_mults} multiplies (tensorProduct+reLinearize, fixed level) at "
                      f"m={shape['M']}, L={l}, K={k}, D={len(shape['digits'])}; CPU restatement of HElib "
                      f"2.2.0 algorithms (not NTL), gcc -O3 -march=native, {dt:.1f} s"}


def cpu_baseline_fresh(cc, sample_mults):
    """The same multLowLvl+reLinearize sequence, host logic over the oracle backend."""
    O = build_native_oracle()
    from helib_amd import ctxt as hc
    from oracle.backend import OKeySwitch, OPoly, OracleOps
    octx = O.Ctx(cc.m)
    for q in cc.primes:
        octx.add_prime(q)
    n = octx.N
    rng = np.random.default_rng(98)
    allp = cc.ctxtPrimes + cc.specialPrimes
    D = len(cc.digits)
    kb = np.stack([uniform_rows(rng, cc.primes, allp, 1, n)[:, 0] for _ in range(D)])
    ka = np.stack([uniform_rows(rng, cc.primes, allp, 1, n)[:, 0] for _ in range(D)])
    W = OKeySwitch(allp, kb, ka)
    ops = OracleOps(octx)
    base = [OPoly(octx, cc.ctxtPrimes, uniform_rows(rng, cc.primes, cc.ctxtPrimes, 1, n)[:, 0])
            for _ in range(4)]
    hc.Ctxt.measure = False   # the oracle's long-double norms would only slow the CPU side down
    fa = hc.Ctxt.fresh(cc, ops, base[0], base[1], ksw=W)
    fb = hc.Ctxt.fresh(cc, ops, base[2], base[3], ksw=W)

    def one():
        a, b = fa.clone(), fb.clone()
        t0 = time.perf_counter()
        a.multLowLvl(b, destructive=True)
        a.reLinearize()
        return time.perf_counter() - t0

    one()
    dt = sum(one() for _ in range(sample_mults))
    return {"value": sample_mults / dt, "unit": "mult/s", "cores": 1, "kind": "port",
            "sample": f"{sample_mults} fresh-ciphertext multiplyBy (bringToSet x2, tensorProduct, "
                      f"dropSmallAndSpecialPrimes, reLinearize; noise bounds, no norm FFTs) at m={cc.m}, "
                      f"L={len(cc.ctxtPrimes)}, K={len(cc.specialPrimes)}; CPU restatement of "
                      f"HElib 2.2.0 algorithms (not NTL), gcc -O3 -march=native, {dt:.1f} s"}


def cpu_baseline_all_cores(bits, sample_mults):
    """the same CPU restatement on every host core at once -- one process
    per core, each timing its own fresh multiplies (independent ciphertexts: the batch axis), rates
    summed.  Separate python processes (no fork of a process that holds a HIP context)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    try:
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(sample_mults),
                                   "--bits", str(bits)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                  text=True) for _ in range(cores)]
        rates = []
        for pr in procs:
            out, _ = pr.communicate(timeout=600)
            rates.append(json.loads(out.strip().splitlines()[-1])["value"])
        return {"value": sum(rates), "unit": "mult/s", "cores": cores,
                "sample": f"{cores} processes x {sample_mults} fresh multiplies each, rates summed, "
                          f"{time.perf_counter() - t0:.1f} s wall"}
    except Exception as e:   # the single-core figure stands on its own
        return {"value": None, "cores": cores, "error": str(e)[:200]}


def run_fixed(hx, ctx, primes, shape, B, steps, warmup, rng, sync, barrier):
    l, k = shape["L"], shape["K"]
    n = ctx.phim
    own, sp = list(range(l)), list(range(l, l + k))
    allp = own + sp
    kb = np.stack([uniform_rows(rng, primes, allp, 1, n)[:, 0] for _ in shape["digits"]])
    ka = np.stack([uniform_rows(rng, primes, allp, 1, n)[:, 0] for _ in shape["digits"]])
    W = hx.KeySwitch(ctx, allp, kb, ka)
    polys = [hx.DoubleCRT(ctx, own, B, uniform_rows(rng, primes, own, B, n)) for _ in range(4)]
    out0, out1 = hx.DoubleCRT(ctx, allp, B), hx.DoubleCRT(ctx, allp, B)

    def step():
        hx.multiplyBy(*polys, W, shape["digits"], out0, out1)

    for _ in range(warmup):
        step()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    barrier()
    return time.perf_counter() - t0


def real_inputs(hx, hc, cc, ctx, B, seed):
    """What benchmarks/bgv_basic.cpp:144-157 prepares: a key pair with its relinearisation matrix and
    two fresh public-key encryptions of random plaintexts -- B independent pairs, generated with
    helib_amd.keys on the device and packed along the batch axis.  Returns the two batched
    ciphertexts and a checker for element b of a product."""
    from helib_amd import keys as hk
    sk = hk.SecKey(cc, hk.HxBackend(ctx, cc), seed)
    sk.GenSecKey(maxDegKswitch=2)
    rng = np.random.default_rng(seed + 1)
    p, n, L = cc.ptxtSpace, cc.phim, len(cc.ctxtPrimes)
    msgs = rng.integers(0, p, size=(2, B, n))
    rows = np.empty((2, 2, L, B, n), dtype=np.uint64)
    first = None
    for j in range(2):
        for b in range(B):
            ct = sk.Encrypt(msgs[j, b])
            first = first or ct
            assert ct.parts["1"].getIndexSet() == list(cc.ctxtPrimes)
            rows[j, 0, :, b] = ct.parts["1"].download()[:, 0]
            rows[j, 1, :, b] = ct.parts["s"].download()[:, 0]
    out = []
    for j in range(2):
        c = first.clone()
        c.parts = {"1": hx.DoubleCRT(ctx, cc.ctxtPrimes, B, rows[j, 0]),
                   "s": hx.DoubleCRT(ctx, cc.ctxtPrimes, B, rows[j, 1])}
        out.append(c)

    def check(result, elements=None):
        """decrypt(result[b]) == m_a[b] * m_b[b] mod (Phi_m, p) for every batch element b (m a power
        of two: X^n + 1).  Returns the number of elements checked; a mismatch aborts the bench."""
        host = {h: (q.getIndexSet(), q.download()) for h, q in result.parts.items()}
        todo = range(B) if elements is None else elements
        for b in todo:
            one = result.clone()
            one.parts = {h: hx.DoubleCRT(ctx, idx, 1, rows_[:, b:b + 1]) for h, (idx, rows_) in host.items()}
            got = sk.Decrypt(one)
            full = np.convolve(msgs[0, b].astype(np.int64), msgs[1, b].astype(np.int64))
            want = (full[:n] - np.append(full[n:], 0)) % p
            if got != [int(v) for v in want]:
                raise SystemExit(f"bench: decrypt(multiplyBy(a, b)) != a*b at batch element {b} -- results are wrong")
        return len(todo)
    return out[0], out[1], check, sk, msgs


def run_fresh(hx, hc, cc, ctx, B, steps, warmup, rng, sync, barrier, measure=True, inputs="real", seed=7,
              prepared=None, mults_per_step=32, verify=True):
    """measure=True: added noise measured as in the reference's default build (canonical-embedding
    norms of the mod-switch deltas and of the key-switch digits, evaluated on the device);
    False: the reference's alternative high-probability bounds, no norms.
    inputs="real": keys and ciphertexts from helib_amd.keys, every element of the last product is
    decrypted and checked against the plaintext product; "uniform": uniform rows (no keys).
    Returns (seconds, result primes, host enqueue seconds, elements verified)."""
    hc.Ctxt.measure = measure
    n = ctx.phim
    allp = cc.ctxtPrimes + cc.specialPrimes
    D = len(cc.digits)
    check = None
    if inputs == "real":
        fa, fb, check = (prepared or real_inputs(hx, hc, cc, ctx, B, seed))[:3]
    else:
        kb = np.stack([uniform_rows(rng, cc.primes, allp, 1, n)[:, 0] for _ in range(D)])
        ka = np.stack([uniform_rows(rng, cc.primes, allp, 1, n)[:, 0] for _ in range(D)])
        W = hx.KeySwitch(ctx, allp, kb, ka)
        base = [hx.DoubleCRT(ctx, cc.ctxtPrimes, B, uniform_rows(rng, cc.primes, cc.ctxtPrimes, B, n))
                for _ in range(4)]
        fa = hc.Ctxt.fresh(cc, hx, base[0], base[1], ksw=W)
        fb = hc.Ctxt.fresh(cc, hx, base[2], base[3], ksw=W)

    def run(k):
        """k x [copy(ctxt1); copy.multiplyBy(ctxt2)], back to back.  Each result's (lazily read) noise
        estimate is completed one multiply later, so that the host has the next multiply queued
        while it waits; a result is dropped once the next one is complete."""
        prev = None
        for _ in range(k):
            a = fa.clone()            # benchmarks/bgv_basic.cpp:160 (untimed there, timed here)
            a.multiplyBy(fb)          # multLowLvl (copies `other`, both bringToSet, tensor) + reLinearize
            if prev is not None:
                _ = prev.lnNoise
            prev = a
            del a
        _ = prev.lnNoise
        return prev

    run(max(1, warmup))
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    last = None
    host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        last = run(mults_per_step)
        host += time.perf_counter() - h0
    sync()
    barrier()
    total = time.perf_counter() - t0
    # host cost of enqueueing one multiply, from an idle device (inside the loop above the HIP
    # queue fills up and the host is throttled to the device's pace, so `host` is not it)
    t1 = time.perf_counter()
    run(4)
    enq = (time.perf_counter() - t1) / 4
    sync()
    nver = check(last) if (check and verify) else 0
    return total, sorted(last.primeSet), host, nver, enq


def batch1_latency(hx, hc, fa, fb, sync, reps=12):
    """One ciphertext pair at a time, as benchmarks/bgv_basic.cpp:158-164 times it: copy (untimed),
    multiplyBy, wait for the device.  Median / minimum of `reps` runs in milliseconds."""
    def one(ct):
        c = ct.clone()
        c.parts = {h: hx.DoubleCRT(q.context, q.getIndexSet(), 1, q.download()[:, 0:1]) for h, q in ct.parts.items()}
        return c
    a1, b1 = one(fa), one(fb)
    ts = []
    for i in range(reps + 2):
        a = a1.clone()
        sync()
        t0 = time.perf_counter()
        a.multiplyBy(b1)
        _ = a.lnNoise
        sync()
        if i >= 2:
            ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    # the same multiply recorded once into a HIP graph and replayed with one launch (the reference's
    # loop multiplies the same two ciphertexts every iteration too); noise bounds instead of measured
    # noise: a recording cannot contain the read-backs
    graph_ms = None
    measure = hc.Ctxt.measure
    try:
        hc.Ctxt.measure = False
        w = a1.clone()
        w.multiplyBy(b1)                      # eagerly once with bounds (plans of this variant)
        sync()
        ctx = a1.parts["1"].context
        ctx.graphBegin()
        rec = a1.clone()
        rec.multiplyBy(b1)
        graph = ctx.graphEnd()
        gs = []
        for i in range(reps + 2):
            sync()
            t0 = time.perf_counter()
            graph.launch()
            sync()
            if i >= 2:
                gs.append((time.perf_counter() - t0) * 1e3)
        gs.sort()
        same = all(np.array_equal(rec.parts[h].download(), w.parts[h].download()) for h in w.parts)
        graph.destroy()
        graph_ms = (gs[len(gs) // 2], gs[0]) if same else "replayed result differs from the eager one"
    except Exception as e:
        graph_ms = f"unavailable: {type(e).__name__}: {str(e)[:120]}"
    finally:
        hc.Ctxt.measure = measure
    return ts[len(ts) // 2], ts[0], graph_ms


def bgv_basic_ops(hx, hc, cc, fa, fb, sk, msgs, sync, reps=8):
    """The other lines of the reference's benchmark list (benchmarks/bgv_basic.cpp:36-211) on the same
    key pair and fresh ciphertexts: += / -= / negate / square / multLowLvl (no relinearisation) /
    rotate by one generator step (EncryptedArray::rotate on a native dimension is one smartAutomorph;
    the key-switching matrix for it is generated here, untimed) over the resident batch -- the copy of
    the operand is made before the timer starts, as the reference pauses its timer for it -- and
    PubKey::Encrypt / SecKey::Decrypt of one ciphertext.  Milliseconds per call and, for the batched
    ones, operations per second; decrypt(op(a, b)) is checked for one batch element of each."""
    from helib_amd import hostnt
    p, n, B = cc.ptxtSpace, cc.phim, fa.parts["1"].batch
    g = hostnt.ZmStar(cc.m, cc.p).gens[0]
    sk.GenKeySWmatrix(1, g)
    fa.ksw_auto[g] = sk.keySwitching[(1, g)].W      # (a ciphertext carries the matrices its key had at encryption)
    ma, mb = msgs[0, 0].astype(object), msgs[1, 0].astype(object)

    def negacyclic(x, y):
        full = np.convolve(np.array(x, dtype=np.int64) % p, np.array(y, dtype=np.int64) % p)
        return [int(v) % p for v in (full[:n] - np.append(full[n:], 0))]

    def rot(x):   # f(X) -> f(X^g) mod X^n + 1
        out = [0] * n
        for i, v in enumerate(x):
            e = i * g % (2 * n)
            out[e % n] = (out[e % n] + (int(v) if e < n else -int(v))) % p
        return out
    ops = {
        "adding_two_ciphertexts": (lambda c: c.__iadd__(fb), lambda: [int(v) % p for v in (ma + mb)]),
        "subtracting_two_ciphertexts": (lambda c: c.__isub__(fb), lambda: [int(v) % p for v in (ma - mb)]),
        "negating_a_ciphertext": (lambda c: c.negate(), lambda: [int(-v) % p for v in ma]),
        "square_a_ciphertext": (lambda c: c.square(), lambda: negacyclic(ma, ma)),
        "multiplying_two_ciphertexts_no_relin": (lambda c: c.multLowLvl(fb), lambda: negacyclic(ma, mb)),
        "rotate_a_ciphertext_by1": (lambda c: c.smartAutomorph(g), lambda: rot(ma)),
    }
    out = {}
    for name, (fn, want) in ops.items():
        copies = [fa.clone() for _ in range(reps + 1)]
        fn(copies[0])                      # warm (plans, slabs)
        _ = copies[0].lnNoise
        sync()
        t0 = time.perf_counter()
        for c in copies[1:]:
            fn(c)
        _ = copies[-1].lnNoise
        sync()
        ms = (time.perf_counter() - t0) / reps * 1e3
        res = copies[-1]
        one = res.clone()
        one.parts = {h: hx.DoubleCRT(q.context, q.getIndexSet(), 1, q.download()[:, 0:1]) for h, q in res.parts.items()}
        if sk.Decrypt(one) != want():
            raise SystemExit(f"bench: decrypt({name}) is wrong")
        out[name] = {"ms_per_call_batch": round(ms, 4), "batch": B, "per_s": round(B / (ms * 1e-3), 1)}
        del copies, res, one
    msg = msgs[0, 0]
    sk.Encrypt(msg)
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ct = sk.Encrypt(msg)
    sync()
    out["encrypting_ciphertexts"] = {"ms_per_call": round((time.perf_counter() - t0) / reps * 1e3, 4), "batch": 1}
    t0 = time.perf_counter()
    for _ in range(reps):
        dec = sk.Decrypt(ct)
    out["decrypting_ciphertexts"] = {"ms_per_call": round((time.perf_counter() - t0) / reps * 1e3, 4), "batch": 1}
    if dec != [int(v) for v in msg]:
        raise SystemExit("bench: decrypt(encrypt(m)) != m")
    out["note"] = ("benchmarks/bgv_basic.cpp:36-211 at m=32768 p=65537; operand copy before the timer as there; "
                   "batched lines run the whole resident batch per call, encrypt/decrypt one ciphertext")
    return out


def ckks_basic_ops(hx, hc, device, stream, sync, bits, B=16, reps=6, m=65536, backend=None, precision=1):
    """The reference's CKKS benchmark list (benchmarks/ckks_basic.cpp:38-236: add / subtract / negate / square /
    rotate by one / multiply without and with relinearisation / encrypt / decrypt / multiply-and-add) at the
    BASELINE configs[3] shape, ContextBuilder<CKKS>().m(65536).precision(1).bits(1400).scale(10): a key pair, B pairs of
    CKKSencrypt-ed ciphertexts packed along the batch axis (python mirror of the host, helib_amd/ctxt.py + keys.py:
    this leg is about the device operations, the timed multiply of the headline runs in the C++ host), the operand
    copy made before the timer as the reference pauses its timer for it.  Every result is decrypted and decoded
    for one batch element and compared with the plaintext operation within the error bound the ciphertext reports."""
    import math
    from helib_amd import hostnt, keys as hk
    cc = hc.ChainContext(m, -1, precision, bits=bits, c=3, ckks=True)
    if backend is None:
        ctx = hx.Context(cc.m, device)
        for q in cc.primes:
            ctx.add_prime(q)
        ctx.set_stream(stream)
        ctx.reserve(12 << 30)         # the copies and results of the timed loops: no hipMalloc inside them
        be = hk.HxBackend(ctx, cc)

        def make(idx, rows_):
            return hx.DoubleCRT(ctx, idx, rows_.shape[1], rows_)
    else:
        be, make = backend(cc)        # (tests: the same list over the CPU checker at a small m)
    sk = hk.SecKey(cc, be, 23)
    sk.GenSecKey(maxDegKswitch=2)
    g = hostnt.ZmStar(cc.m, cc.p).gens[0]
    sk.GenKeySWmatrix(1, g)
    n, L = cc.phim, len(cc.ctxtPrimes)
    # the factor PubKey::Encrypt(Ptxt<CKKS>) encodes with and plaintexts whose canonical embedding stays below the
    # declared size 1 (as helib_amd/csrc/host_session.cpp)
    f = float(cc.encodeScalingFactor())
    rng = np.random.default_rng(29)
    vals = rng.uniform(-1, 1, size=(2, B, n)) / (8.0 * math.sqrt(n / 3.0))
    enc = np.rint(vals * f) / f                                   # what is actually encrypted
    rows = np.empty((2, 2, L, B, n), dtype=np.uint64)
    first = None
    for j in range(2):
        for b in range(B):
            ct = sk.CKKSencrypt(np.rint(vals[j, b] * f).astype(np.int64), 1.0, f)
            first = first or ct
            rows[j, 0, :, b] = np.asarray(ct.parts["1"].download()).reshape(L, -1, n)[:, 0]
            rows[j, 1, :, b] = np.asarray(ct.parts["s"].download()).reshape(L, -1, n)[:, 0]
    ops_in = []
    for j in range(2):
        c = first.clone()
        c.parts = {"1": make(list(cc.ctxtPrimes), rows[j, 0]), "s": make(list(cc.ctxtPrimes), rows[j, 1])}
        ops_in.append(c)
    fa, fb = ops_in
    fa.ksw_auto[g] = sk.keySwitching[(1, g)].W

    def decode(ct, b=0):
        one = ct.clone()
        one.parts = {h: make(q.getIndexSet(), np.asarray(q.download()).reshape(len(q.getIndexSet()), -1, n)[:, b:b + 1])
                     for h, q in ct.parts.items()}
        return np.array([float(v) for v in sk.Decrypt(one)]) / math.exp(ct.lnRatFactor)

    def nega(x, y):
        full = np.convolve(x, y)
        return full[:n] - np.append(full[n:], 0.0)

    def rot(x):                                                   # f(X) -> f(X^g) mod X^n + 1
        out = np.zeros(n)
        e = (np.arange(n) * g) % (2 * n)
        np.add.at(out, e % n, np.where(e < n, x, -x))
        return out

    def mul_add(c):
        c.multiplyBy(fb)
        c += fb          # (the reference adds into ctxt1; the same device work: operator+= equalises level and scale first)
    a0, b0 = enc[0, 0], enc[1, 0]
    ops = {
        "adding_two_ciphertexts": (lambda c: c.__iadd__(fb), lambda: a0 + b0),
        "subtracting_two_ciphertexts": (lambda c: c.__isub__(fb), lambda: a0 - b0),
        "negating_a_ciphertext": (lambda c: c.negate(), lambda: -a0),
        "square_a_ciphertext": (lambda c: c.square(), lambda: nega(a0, a0)),
        "rotate_a_ciphertext_by1": (lambda c: c.smartAutomorph(g), lambda: rot(a0)),
        "multiplying_two_ciphertexts_no_relin": (lambda c: c.multLowLvl(fb), lambda: nega(a0, b0)),
        "multiplying_two_ciphertexts": (lambda c: c.multiplyBy(fb), lambda: nega(a0, b0)),
        "multiply_and_add_two_ciphertexts": (mul_add, lambda: nega(a0, b0) + b0),
    }
    # burn-in: the HIP runtime grows its own command / signal pools the first time a process keeps this much work in
    # flight, a one-time event of tens of milliseconds in a process that already maps tens of GB -- it landed in whichever
    # op happened to be the fourth or fifth heavy call (profiles/r03_bench_line_ckks65536_*: square or rotate at 10 ms)
    for wc in [fa.clone() for _ in range(12)]:
        wc.multiplyBy(fb)
    for wc in [fa.clone() for _ in range(6)]:
        wc.smartAutomorph(g)
    _ = wc.lnNoise
    del wc
    sync()
    out = {}
    for name, (fn, want) in ops.items():
        try:
            # warm-up: three calls, enqueued as deep as the timed loop will be -- plans and slabs, and the HIP runtime's
            # own command / signal pools, whose growth is a one-time event of tens of milliseconds in this process
            # (it maps tens of GB already) and would otherwise land in one of the few timed calls
            for wc in [fa.clone() for _ in range(3)]:
                fn(wc)
            _ = wc.lnNoise
            del wc
            copies = [fa.clone() for _ in range(reps + 1)]
            sync()
            prof = None
            if os.environ.get("HX_BENCH_PYPROFILE") == name:      # debugging aid: where the host time of one line goes
                import cProfile
                prof = cProfile.Profile()
                prof.enable()
            gprof = backend is None and os.environ.get("HX_BENCH_GPUPROFILE") == name
            if gprof:
                hx.profileBegin()
            t0 = time.perf_counter()
            each = []
            for c in copies[1:]:
                e0 = time.perf_counter()
                fn(c)
                each.append(round((time.perf_counter() - e0) * 1e3, 3))
            e0 = time.perf_counter()
            _ = copies[-1].lnNoise
            each.append(round((time.perf_counter() - e0) * 1e3, 3))
            th = time.perf_counter()
            if gprof:
                sync()
                for kk in hx.profileEnd()["kernels"][:12]:
                    print("  [gpu profile]", name, kk["kernel"][:50], kk["workgroups"], kk["calls"], round(kk["avg_us"], 1),
                          round(kk["max_us"], 1), file=sys.stderr)
            if prof:
                import pstats
                prof.disable()
                pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(25)
            sync()
            ms = (time.perf_counter() - t0) / reps * 1e3
            host_ms = (th - t0) / reps * 1e3
            res, w = copies[-1], want()
            err = float(np.max(np.abs(decode(res) - w)))
            tol = math.exp(res.lnNoise - res.lnRatFactor)
        except SystemExit:
            raise
        except Exception as e:
            out[name] = f"unavailable: {type(e).__name__}: {str(e)[:120]}"
            continue
        # (precision(r) promises 2^-r: from ten bits on the result must be right to 1e-3 of its size; at precision(1)
        # the reported bound exceeds the values themselves, so the result must also CORRELATE with the expected one
        # beyond 8 standard deviations of what an unrelated result would show -- helib_amd/host.py: Session.verify)
        from helib_amd import host as hh
        corr = hh.ckks_correlation(decode(res), w)
        if not (err <= tol and (err <= 1e-3 * max(float(np.max(np.abs(w))), 1e-30) if precision >= 10
                                else corr >= 8.0 / math.sqrt(len(w)))):
            raise SystemExit(f"bench: decode(decrypt({name})) is off by {err:g} (reported bound {tol:g}, correlation {corr:.4f})")
        out[name] = {"ms_per_call_batch": round(ms, 4), "host_ms_per_call": round(host_ms, 4), "host_ms_each_call_then_norm_wait": each,
                     "batch": B,
                     "per_s": round(B / (ms * 1e-3), 1),
                     "decode_max_abs_err": float(f"{err:.3g}"), "reported_error_bound": float(f"{tol:.3g}"),
                     "correlation_with_expected": round(corr, 4)}
        del copies, res
    msg = np.rint(vals[0, 0] * f).astype(np.int64)
    sk.CKKSencrypt(msg, 1.0, f)
    sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ct = sk.CKKSencrypt(msg, 1.0, f)
    sync()
    out["encrypting_ciphertexts"] = {"ms_per_call": round((time.perf_counter() - t0) / reps * 1e3, 4), "batch": 1}
    t0 = time.perf_counter()
    for _ in range(reps):
        dec = sk.Decrypt(ct)
    out["decrypting_ciphertexts"] = {"ms_per_call": round((time.perf_counter() - t0) / reps * 1e3, 4), "batch": 1}
    got = np.array([float(v) for v in dec]) / math.exp(ct.lnRatFactor)
    if float(np.max(np.abs(got - enc[0, 0]))) > math.exp(ct.lnNoise - ct.lnRatFactor):
        raise SystemExit("bench: decode(decrypt(CKKSencrypt(v))) is off")
    out["note"] = (f"benchmarks/ckks_basic.cpp:38-236 at m={m} precision={precision} bits={bits} (L={L}, K={len(cc.specialPrimes)}); "
                   "operand copy before the timer as there; batched lines run the whole batch per call, encrypt / decrypt one "
                   "ciphertext; python mirror of the host")
    return out


def run_session(sess, level, steps, warmup, R, sync, barrier, measure):
    """The timed region, driven by the C++ host: `steps` x hxh_multiply(level, R) -- R x [copy(a);
    copy.multiplyBy(b)] enqueued back to back by helib_amd/csrc/host_session.cpp (level 1: the two fresh
    ciphertexts; level 2: the kept level-1 product with itself).  Returns (seconds, host seconds)."""
    for _ in range(warmup):
        sess.multiply(level, R, measure)
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(steps):
        h0 = time.perf_counter()
        sess.multiply(level, R, measure)
        host += time.perf_counter() - h0
    sync()
    barrier()
    return time.perf_counter() - t0, host


def levels_leg(hh, sparams, batch, R, device, stream, sync, warm_steps=2, steps=5, elements=None, seed=17, hx=None):
    """A second workload through the same C++ host: level 1 (fresh x fresh) and level 2 (product x product) of
    `sparams` = (scheme, m, p, r, bits), `steps` x R multiplies of a `batch`-pair batch each after `warm_steps` full
    steps (plans, arena chunks and clocks settle there; the session reserved its working set at creation), noise
    measured.  hipMalloc calls made inside the two timed windows are reported (must be 0: an allocation in a timed
    window once doubled a figure of this leg)."""
    t0 = time.perf_counter()
    so = hh.Session(*sparams, batch, device=device, stream=stream, seed=seed)
    sync()
    setup = time.perf_counter() - t0
    out = {"workload": f"{sparams[0].upper()} m={sparams[1]} " + (f"precision={sparams[3]}" if sparams[0] == "ckks" else f"p={sparams[2]}")
           + f" bits={sparams[4]}: L={so.L_ctxt}x{so.ctxt_bits}b, K={so.K}x{so.special_bits}b, D={so.D}, batch {so.batch}, "
           f"{steps} x {R} multiplies timed per level after {warm_steps} x {R}, noise measured, C++ host",
           "setup_s": round(setup, 2)}
    mallocs = 0
    for level in (1, 2):
        run_session(so, level, warm_steps, 1, R, sync, lambda: None, True)
        m0 = so.arena_stats()["hipMalloc_calls"]
        dt, _ = run_session(so, level, steps, 0, R, sync, lambda: None, True)
        mallocs += so.arena_stats()["hipMalloc_calls"] - m0
        nv = so.verify(level, elements=elements)          # every batch element (16 .. 128 of them) unless told otherwise
        out[f"level{level}_mult_per_s"] = round(so.batch * R * steps / dt, 1)
        if hx is not None:
            # this leg's own whole-operation fraction of the 8 TB/s roofline: the compulsory bytes of the kernels that
            # ran (kernel_table's algorithmic bytes per launch x launches per multiply) against the measured rate
            try:
                prof = in_situ_profile(hx, so, level, 2, sync, warm=4)
                table, _ = kernel_table(prof, so.phim, so.batch, so.L_ctxt, so.K, so.D, 2)
                fused = sum(r_["algorithmic_bytes_per_launch"] * r_["launches_per_multiply"] for r_ in table
                            if "algorithmic_bytes_per_launch" in r_) / so.batch
                share = sum(r_["share"] for r_ in table if "algorithmic_bytes_per_launch" in r_)
                out[f"level{level}_fused_algorithmic_MB_per_mult"] = round(fused / 1e6, 2)
                out[f"level{level}_value_over_fused_roofline"] = round(out[f"level{level}_mult_per_s"] / (HBM_PEAK_GBS * 1e9 / fused), 4)
                out[f"level{level}_fused_bytes_cover_share_of_kernel_time"] = round(share, 3)
            except Exception as e:       # (a leg whose kernels have no byte model: the rate stands alone)
                out[f"level{level}_value_over_fused_roofline"] = f"unavailable: {type(e).__name__}: {str(e)[:120]}"
        out[f"level{level}_ms_per_mult_of_the_batch"] = round(dt / (steps * R) * 1e3, 4)
        out[f"level{level}_verified_elements"] = nv
        out[f"level{level}_result_primes"] = len(so.result_primes(level))
    out["level2_over_level1"] = round(out["level2_ms_per_mult_of_the_batch"] / out["level1_ms_per_mult_of_the_batch"], 3)
    out["hipMalloc_calls_in_timed_windows"] = mallocs
    if sparams[0] == "ckks":   # (the verified elements cleared this: Session.verify)
        out["verified_min_correlation_with_expected_product"] = round(getattr(so, "min_ckks_correlation", float("nan")), 4)
    return out, so


# static instruction counts of the row-transform kernels per 512-thread workgroup (VALU
# instructions per wave and row x 8 waves x 64 lanes) and their 64-bit modular multiplications per row
# (round 5, the Proth-form path: DYNAMIC counts per wave from the SQ_INSTS_VALU pass, profiles/r05_valu_floor_by_class.json)
VALU_PER_WAVE = {"ntt_row_kernel<14, false, 8>": 3903, "ntt_moddown_apply_kernel<14, false>": 4836,
                 "ntt_moddown_apply_tensor_kernel<14, false>": 5455}


def kernel_table(prof, n, B, l, k, d, mults):
    """Per-kernel view of an in-situ profile (hx_profile_end) of `mults` multiplies of the batch: time per
    multiply, share, and -- where SURVEY 8(d) / DESIGN.md section 3 define them -- algorithmic bytes per
    launch, achieved GB/s and the fraction of the 8 TB/s peak.  n = phi(m), l / k / d = ctxt primes /
    special primes / digits of the ciphertext being multiplied."""
    total = sum(kk["total_us"] for kk in prof["kernels"]) or 1.0
    logn = n.bit_length() - 1
    rows = []
    for kk in prof["kernels"]:
        name, w = kk["kernel"].replace("hx::", ""), kk["workgroups"]
        byts, what = None, None
        if name.startswith("ntt_row_kernel<") or name.startswith("ntt_moddown_prep"):
            byts, what = 16 * n * w, "16N per row"
        elif name.startswith("ntt_moddown_apply_kernel<") and name.endswith("false>"):
            nk = l                                   # kept rows of a single-prime bringToSet at this level
            el = max(1, w // nk)
            # per element: c_r in + out on every kept row (the row the fused mod-up adds has no input), x and S once
            byts, what = el * (nk * 16 * n - 8 * n + 16 * n), f"{el} elements x ({nk} kept rows x 16N - 8N (added row) + x,S 16N)"
        elif name.startswith("ntt_moddown_apply_kernel<"):
            byts, what = w * 24 * n, "delta/P in 8N + c_r in + out 16N per kept row"
        elif name.startswith("ntt_moddown_apply_tensor_kernel<") and name.endswith("true>"):
            # tensorProduct folded into the product's SEVERAL-primes mod-switch (every multiply after the first):
            # nk kept rows per part, of which the k special primes are added by the fused mod-up (no operand rows)
            nk = max(1, w // (3 * B))
            nin = nk - k if nk > k else nk
            byts = B * 8 * n * (3 * nk + 4 * nin + 3 * nk)
            what = f"{B} elements x 8N x (3 parts x {nk} delta rows in + 4 operand parts x {nin} rows in + 3 parts x {nk} rows out)"
        elif name.startswith("ntt_moddown_apply_tensor_kernel<"):
            nk = l                                   # tensorProduct folded into the product's single-prime mod-switch
            el = max(1, w // (3 * nk))               # batch elements (three product parts each)
            # compulsory traffic per batch element: the four operand parts' nk-1 real rows in once (the kernel reads
            # a row once per product part that uses it: 8 row reads for 4 rows), the three parts' nk rows out, x and S
            byts = el * 8 * n * (4 * (nk - 1) + 3 * nk + 6)
            what = f"{el} elements x 8N x (4 operand parts x {nk - 1} rows in + 3 parts x {nk} rows out + x,S of 3 parts)"
        elif name.startswith("ntt_moddown_prep_multi_tensor_kernel<"):
            byts, what = (w // 3) * 8 * n * (4 + 3), "4 operand rows in, x of 3 parts out per batch element and dropped prime"
        elif name.startswith("ntt_moddown_prep_tensor_kernel<"):
            byts, what = (w // 3) * 8 * n * (4 + 3), "4 operand rows in, x of 3 parts out per batch element"
        elif name.startswith("tensor_kernel"):
            byts, what = 56 * n * l * B, "(4 in + 3 out) x 8N per prime row"
        elif name.startswith("keyswitch_kernel"):
            ext = d * (l + k) - l
            byts = 8 * n * (B * (ext + l + 2 * l + 2 * (l + k)) + 2 * d * (l + k))
            what = "extension rows + s^2 rows + parts (1),(s) in, 2(L+K) rows out per element; key rows once per batch"
        elif name.startswith("rns_extend_mfma_kernel<") and d > 0:
            # the basis extension of one digit on the matrix cores (bits = 6400: digits of 36 / 36 / 35 primes onto the
            # other 107 / 108): 128 coefficients per workgroup; n from the kernel's step count (4 steps - 1 >= n + ...)
            steps = int(name[len("rns_extend_mfma_kernel<"):].split(">")[0])
            nsrc = next((c for c in ((l + d - 1) // d, l // d) if (c + 1 + 3) // 4 == steps), 4 * steps - 1)
            ntg = l + k - nsrc
            coefs = min(w * 128, B * n)
            byts, what = coefs * 8 * (nsrc + ntg), f"{nsrc} source rows in, {ntg} target rows out (the later digits' in-place update words not counted)"
            macs = coefs * (32 * steps) * (32 * ((ntg + 3) // 4))
        elif name.startswith("break_digits"):
            byts, what = 8 * n * B * (l + d * (l + k) - l + d), "L rows in, D(L+K)-L extension rows + D fraction rows out"
        elif name.startswith("moddown_S_kernel"):
            byts, what = min(w, 8192) * 256 * 16, "x in, S out"
        elif name.startswith("embed_norm_quarter_kernel<hx::NormSrcXS") or name.startswith("embed_norm_quarter_kernel<NormSrcXS"):
            byts, what = w * 16 * n, "x and S in"
        elif name.startswith("embed_norm_quarter"):
            byts, what = w * 8 * n, "N doubles in"
        row = {"kernel": name, "workgroups": w, "launches_per_multiply": round(kk["calls"] / mults, 3),
               "avg_us": round(kk["avg_us"], 2), "min_us": round(kk["min_us"], 2), "max_us": round(kk["max_us"], 2),
               "us_per_multiply": round(kk["total_us"] / mults, 2), "share": round(kk["total_us"] / total, 4)}
        if byts:
            ach = byts / (kk["avg_us"] * 1e-6) / 1e9
            row.update({"algorithmic_bytes_per_launch": int(byts), "bytes_are": what, "achieved_GBps": round(ach, 1),
                        "frac": round(ach / HBM_PEAK_GBS, 4)})
        if name.startswith("rns_extend_mfma_kernel<") and d > 0:
            # the one kernel of this path with its sums on the matrix cores: int8 multiply-adds of the limb matrix
            # product, against the dense int8 rate the micro-benchmark reaches (tools/ubench/mfma_i8_bench.hip: 3.6 POPS
            # of the guide's >= 3.9; two operations per multiply-add)
            row.update({"int8_macs_per_launch": int(macs), "achieved_int8_TOPS": round(2 * macs / (kk["avg_us"] * 1e-6) / 1e12, 1),
                        "frac_of_dense_int8_peak": round(2 * macs / (kk["avg_us"] * 1e-6) / 1e12 / 3944.0, 4)})
        if name in VALU_PER_WAVE and logn == 14:
            # the integer rate these kernels run at -- VALU lane-operations and 64-bit modular
            # multiplications (Shoup products: N/2 log2 N butterflies, + 2N for the mod-down apply's load / store)
            lane_ops = w * 8 * VALU_PER_WAVE[name] * 64
            mm = w * (n // 2 * logn + (2 * n if "apply" in name else 0))
            row["valu_lane_ops_per_s"] = float(f"{lane_ops / (kk['avg_us'] * 1e-6):.4g}")
            row["modmul64_per_s"] = float(f"{mm / (kk['avg_us'] * 1e-6):.4g}")
        rows.append(row)
    return rows, total / mults


def in_situ_profile(hx, sess, level, mults, sync, warm=24):
    """`mults` multiplies of the batch with every kernel launch carrying the dispatch's own start / stop events.
    A stretch of un-profiled multiplies runs right before (the decryption check in front of this leaves the
    device nearly idle for seconds and its clocks low; 24 multiplies of the batch are ~0.1 s of full load)."""
    sess.multiply(level, warm, True)
    sync()
    hx.profileBegin()
    sess.multiply(level, mults, True)
    sync()
    return hx.profileEnd()


def config5_leg(hx, iters=5, batch=32):
    """BASELINE configs[4]: the general-m transform at m=21845 (phi = 16384; the reference: Bluestein with a 2^16-point
    convolution), DoubleCRT of L=16 primes from PrimeGenerator(60, 21845), forward and inverse transforms of `batch`
    objects (512 rows: the chip is filled), timed with HIP events on the context's stream; algorithmic bytes = 16N per
    row.  Round 6: the engine runs it as Good-Thomas x Rader (21845 = 5 * 17 * 257, pfa_kernels.hip), one
    launch per direction, rem Phi_m fused into the inverse; `method` says which kernels actually ran (HX_NO_PFA=1:
    Bluestein on the convolution kernels, as rounds 3-5)."""
    from helib_amd import hostnt
    m, L = 21845, 16
    g = hostnt.PrimeGen(60, m)
    primes = [g.next() for _ in range(L)]
    ctx = hx.Context(m)
    for q in primes:
        ctx.add_prime(q)
    n = ctx.phim
    rng = np.random.default_rng(5)
    rows = uniform_rows(rng, primes, list(range(L)), batch, n)
    d = hx.DoubleCRT(ctx, list(range(L)), batch, rows)
    d.FFT()
    d.iFFT()
    ok = bool(np.array_equal(d.download(), rows))
    out = {"workload": f"Cmodulus::FFT / iFFT at m={m} phi={n} L={L} batch {batch} ({L * batch} rows); the reference: Bluestein, conv length 2^16",
           "round_trip_exact": ok}
    for name, fn in (("forward", d.FFT), ("inverse", d.iFFT)):
        ctx.timerBegin()
        for _ in range(iters):
            fn()
        ms = ctx.timerEnd() / iters
        byts = 16.0 * n * L * batch
        out[name] = {"ms": round(ms, 4), "ns_per_row": round(ms * 1e6 / (L * batch), 1),
                     "achieved_GBps": round(byts / (ms * 1e-3) / 1e9, 1), "frac": round(byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    hx.profileBegin()
    d.FFT()
    d.iFFT()
    ctx.sync()
    prof = hx.profileEnd()
    tot = sum(kk["total_us"] for kk in prof["kernels"]) or 1.0
    names = " ".join(kk["kernel"] for kk in prof["kernels"])
    out["method"] = ("Good-Thomas x Rader (5 x 17 x 257), rem Phi_m as binomial passes in the same launch" if "pfa_row_kernel<2," in names
                     else "Good-Thomas x Rader, rem Phi_m on the convolution kernels" if "pfa_row_kernel" in names
                     else "Bluestein (chirp convolution)")
    out["kernels_in_situ_fwd_plus_inv"] = [{"kernel": kk["kernel"].replace("hx::", ""), "workgroups": kk["workgroups"],
                                            "calls": kk["calls"], "avg_us": round(kk["avg_us"], 1),
                                            "share": round(kk["total_us"] / tot, 3)} for kk in prof["kernels"][:12]]
    return out


def config1_2_leg(hx):
    """BASELINE configs[0] and [1].  Config 1 = benchmarks/fft_bench.cpp:24-73: Cmodulus forward / inverse transform
    at m=16384 (N=8192) modulo one 49-bit prime -- one row at a time as fft_bench times it (latency, input and
    output resident on the device) and 4096 rows per launch (GB/s by 16N per row).  Config 2 = DoubleCRT += and *=
    at m=32768, L=16 primes of 60 bits (src/DoubleCRT.cpp:216-337), batch 64 and batch 1, GB/s by 24N per row.
    HIP events on the context's stream."""
    from helib_amd import hostnt
    rng = np.random.default_rng(7)
    out = {}

    def timed(ctx, fn, iters):
        fn()
        ctx.timerBegin()
        for _ in range(iters):
            fn()
        return ctx.timerEnd() / iters * 1e-3      # seconds per call
    m = 16384
    q = hostnt.PrimeGen(49, m).next()
    ctx = hx.Context(m)
    ctx.add_prime(q)
    n = ctx.phim
    c1 = {"workload": f"fft_bench: m={m} N={n}, one {q.bit_length()}-bit prime (benchmarks/fft_bench.cpp:24-73)"}
    for B, tag in ((1, "one_row"), (4096, "4096_rows")):
        d = hx.DoubleCRT(ctx, [0], B, rng.integers(0, q, size=(1, B, n), dtype=np.uint64))
        tf, ti = timed(ctx, d.FFT, 30), timed(ctx, d.iFFT, 30)
        c1[f"forward_us_{tag}"] = round(tf * 1e6, 2)
        c1[f"inverse_us_{tag}"] = round(ti * 1e6, 2)
        if B > 1:
            c1["forward_GBps"] = round(16 * n * B / tf / 1e9, 1)
            c1["inverse_GBps"] = round(16 * n * B / ti / 1e9, 1)
            c1["forward_frac"] = round(16 * n * B / tf / 1e9 / HBM_PEAK_GBS, 4)
            c1["inverse_frac"] = round(16 * n * B / ti / 1e9 / HBM_PEAK_GBS, 4)
        del d
    out["config1_fft_bench"] = c1
    del ctx
    m, L = 32768, 16
    g = hostnt.PrimeGen(60, m)
    primes = [g.next() for _ in range(L)]
    ctx = hx.Context(m)
    for p in primes:
        ctx.add_prime(p)
    n = ctx.phim
    idx = list(range(L))
    c2 = {"workload": f"DoubleCRT += / *= at m={m} N={n}, L={L} x 60-bit primes (src/DoubleCRT.cpp:216-337), 24N bytes per row"}
    for B in (64, 1):
        a = hx.DoubleCRT(ctx, idx, B, uniform_rows(rng, primes, idx, B, n))
        b = hx.DoubleCRT(ctx, idx, B, uniform_rows(rng, primes, idx, B, n))
        ta, tm = timed(ctx, lambda: a.__iadd__(b), 60), timed(ctx, lambda: a.__imul__(b), 60)
        byts = 24.0 * n * L * B
        c2[f"batch{B}"] = {"add_us": round(ta * 1e6, 2), "mul_us": round(tm * 1e6, 2), "add_GBps": round(byts / ta / 1e9, 1),
                           "mul_GBps": round(byts / tm / 1e9, 1), "add_frac": round(byts / ta / 1e9 / HBM_PEAK_GBS, 4),
                           "mul_frac": round(byts / tm / 1e9 / HBM_PEAK_GBS, 4)}
        del a, b
    out["config2_doublecrt_add_mul"] = c2
    return out


def ntt_back_to_back(hx, ctx, primes_list, own, sp, digits, B, rng, iters):
    """The forward / inverse row transform alone, `iters` launches back to back at the launch shape the forward
    transform has inside the key switch (D*(L+K)-L rows x B), HIP events on the launch stream.  This is the
    favourable figure (warm tables, nothing else in the caches); the in-situ one is what a multiply is made of."""
    n = ctx.phim
    nrows = len(digits) * (len(own) + len(sp)) - len(own)
    digp = hx.DoubleCRT(ctx, own, B, uniform_rows(rng, primes_list, own, B, n))
    dg = digp.breakIntoDigits(digits, sp)          # D*(L+K) rows x B, evaluation domain
    hx.time_ntt(dg, True, 2, nrows)                 # warm both directions
    hx.time_ntt(dg, False, 2, nrows)
    ms_inv = hx.time_ntt(dg, True, iters, nrows)
    ms_fwd = hx.time_ntt(dg, False, iters, nrows)
    bytes_launch = 16.0 * n * nrows * B             # 16*N bytes per row transform
    return {"rows_per_launch": nrows * B, "bytes_per_launch": bytes_launch,
            "forward_avg_launch_ms": round(ms_fwd, 4), "inverse_avg_launch_ms": round(ms_inv, 4),
            "forward_frac": round(bytes_launch / (ms_fwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "inverse_frac": round(bytes_launch / (ms_inv * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def make_roofline(table, n, B, l, k, d, b2b=None):
    """`roofline` = the time-dominant kernel of the in-situ table at its dominant launch shape."""
    with_bytes = [r for r in table if "frac" in r]
    if not with_bytes:
        return None
    by_kernel = {}
    for r in table:
        by_kernel[r["kernel"]] = by_kernel.get(r["kernel"], 0.0) + r["us_per_multiply"]
    dom_name = max(by_kernel, key=by_kernel.get)
    cands = [r for r in with_bytes if r["kernel"] == dom_name] or with_bytes
    dom = max(cands, key=lambda r: r["us_per_multiply"])
    roof = {"bound": "hbm", "kernel": dom["kernel"], "measured": "in situ: HIP events around every launch of this kernel "
            "inside the multiply sequence (hx_profile_begin/_end), on the stream it is launched on",
            "achieved": dom["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
            "frac_of_copy_ceiling": round(dom["achieved_GBps"] / HBM_COPY_GBS, 4),
            "workgroups_per_launch": dom["workgroups"], "avg_launch_us": dom["avg_us"],
            "bytes_per_launch": dom["algorithmic_bytes_per_launch"], "bytes_are": dom["bytes_are"],
            "share_of_step": round(by_kernel[dom_name] / (sum(by_kernel.values()) or 1.0), 4),
            "other_launch_shapes": [{kk: r[kk] for kk in ("workgroups", "avg_us", "achieved_GBps", "frac")}
                                    for r in cands if r is not dom]}
    for kk in ("valu_lane_ops_per_s", "modmul64_per_s"):
        if kk in dom:
            roof[kk] = dom[kk]
    if "modmul64_per_s" in dom:
        roof["int_ops_per_s"] = dom["modmul64_per_s"]
        roof["int_ops_are"] = ("64-bit modular multiplications (Proth-form Montgomery products: 6 multiply-adds + 2 carries each; "
                               "Shoup products of 9 multiplier instructions on rows of other primes) per "
                               "second over the chip; valu_lane_ops_per_s = static VALU instruction count x 64 lanes / time")
    # recorded PMC traffic of the same kernel and launch shape (cannot be collected inside this process)
    traffic, src = None, None
    short = dom["kernel"].replace(", false>", ">").replace(", true>", ",plain>")
    for rec in ("r06_pmc_roofline_kernel_traffic.json", "r05_pmc_roofline_kernel_traffic.json", "r04_pmc_roofline_kernel_traffic.json", "r03_pmc_roofline_kernel_traffic.json"):
        t, sname = recorded_traffic(rec, short, dom["workgroups"], n)
        if t is not None:
            traffic, src = t, sname
            break
    # the instruction-class-weighted issue floor of this kernel (recorded: tools/valu_floor.py over the kernel's ISA with
    # the per-class issue costs of tools/ubench/issue_bench, profiles/r05_valu_floor_by_class.json: the Proth-form path of the
    # round-5 kernels), scaled to this launch
    try:
        with open(os.path.join(ROOT, "profiles", "r05_valu_floor_by_class.json")) as f:
            fl = json.load(f)["kernels"]
        m = dom["kernel"].replace(" ", "")
        tag = m[:m.index("<")] + "ILi" + m[m.index("<") + 1:m.index(",")] + "ELb" + ("1" if m.endswith("true>") else "0")
        for e in fl:
            if tag in e["kernel"] and "floor_us_of_the_in_situ_launch_w8_dynamic_count" in e:
                fus = e["floor_us_of_the_in_situ_launch_w8_dynamic_count"] * dom["workgroups"] / e["in_situ_launch"]["workgroups"]
                roof["issue_floor"] = {"floor_us_per_launch": round(fus, 1), "measured_over_floor": round(dom["avg_us"] / fus, 3),
                                       "valu_per_wave_dynamic": e["dynamic_valu_per_wave_SQ_INSTS_VALU"],
                                       "kind": "recorded per-class issue costs (idle-chip micro-benchmark) x this kernel's instruction "
                                               "histogram scaled to its SQ_INSTS_VALU; profiles/r05_valu_floor_by_class.json"}
    except Exception:
        pass
    roof["traffic"] = traffic
    roof["traffic_kind"] = "recorded (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same launch shape)" if traffic else None
    roof["traffic_source"] = src
    fwd = [r for r in with_bytes if r["kernel"].startswith("ntt_row_kernel<") and ", false" in r["kernel"]]
    inv = [r for r in with_bytes if r["kernel"].startswith("ntt_row_kernel<") and ", true" in r["kernel"]]
    if fwd:
        f = max(fwd, key=lambda r: r["us_per_multiply"])
        roof["ntt_forward"] = {"kernel": f["kernel"], "workgroups_per_launch": f["workgroups"], "avg_launch_us_in_situ": f["avg_us"],
                               "frac_in_situ": f["frac"]}
        if b2b and b2b["rows_per_launch"] == f["workgroups"]:
            roof["ntt_forward"].update({"avg_launch_us_back_to_back": round(b2b["forward_avg_launch_ms"] * 1e3, 2),
                                        "frac_back_to_back": b2b["forward_frac"]})
        for kk in ("valu_lane_ops_per_s", "modmul64_per_s"):
            if kk in f:
                roof["ntt_forward"][kk] = f[kk]
    if inv:
        f = max(inv, key=lambda r: r["us_per_multiply"])
        roof["ntt_inverse"] = {"kernel": f["kernel"], "workgroups_per_launch": f["workgroups"], "avg_launch_us_in_situ": f["avg_us"],
                               "frac_in_situ": f["frac"]}
        if b2b:
            roof["ntt_inverse"]["frac_back_to_back_at_the_forward_shape"] = b2b["inverse_frac"]
    return roof


def recorded_traffic(name, kernel, rows, n):
    """HBM bytes per launch from the PMC counters cannot be collected inside this process: they come
    from the committed rocprofv3 --pmc passes over the same kernel and launch shape (separate passes,
    FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md, calibrated on kernels of known
    traffic in the same passes) -- None when no recorded pass matches this launch."""
    rec = os.path.join(ROOT, "profiles", name)
    if os.path.exists(rec):
        with open(rec) as f:
            recs = json.load(f)
        for r in recs if isinstance(recs, list) else [recs]:
            if r.get("kernel") == kernel and r.get("rows_per_launch") == rows and r.get("N") == n:
                return r["traffic_bytes_per_launch"], "profiles/" + name
    return None, None


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n, argv):
    """`bench.py --gpus N` started as ONE process: spawn the N ranks (one process per GPU), pass
    rank 0's JSON line through, fail if any rank fails."""
    port = free_port()
    procs, logs = [], []
    logdir = os.environ.get("HX_RANK_LOG_DIR") or tempfile.mkdtemp(prefix="helib_amd_ranks_")
    os.makedirs(logdir, exist_ok=True)
    for r in range(n):
        env = dict(os.environ, WORLD_SIZE=str(n), RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # ranks > 0 keep what they print (and their stderr) in a per-rank file: a rank that dies says why
        log = open(os.path.join(logdir, f"rank{r}.log"), "w") if r else None
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else log, stderr=None if r == 0 else subprocess.STDOUT,
                                      text=True))
    # watchdog: every rank is polled; the first one that exits with an error ends the others (a rank that dies during
    # setup would otherwise leave the rest inside a collective until RCCL's own timeout: ten minutes of an 8-GPU lease)
    # and its log is what the launcher prints.  Rank 0's stdout is drained on a thread so that it never blocks on a full pipe.
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    first_bad = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad and first_bad is None:
            first_bad = bad[0]
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.terminate()
            deadline = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
        if all(rc is not None for rc in [p.poll() for p in procs]):
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=10)
    out = "".join(c for c in chunks if c)
    for log in logs:
        if log:
            log.close()
    sys.stdout.write(out)
    sys.stdout.flush()
    if any(rcs):
        if first_bad is not None:
            sys.stderr.write(f"bench.py: rank {first_bad} failed first (exit code {rcs[first_bad]}); the other ranks were stopped\n")
        for r, rc in enumerate(rcs):
            if r and rc and (first_bad in (None, 0) or r == first_bad):
                try:
                    with open(os.path.join(logdir, f"rank{r}.log")) as f:
                        sys.stderr.write(f"---- rank {r} (exit code {rc}), last lines of {f.name}:\n" + "".join(f.readlines()[-15:]))
                except OSError:
                    pass
        raise SystemExit(f"bench.py: rank exit codes {rcs} (per-rank logs in {logdir})")


def dry_rank(args):
    """--dry-launch: the launch / shard / barrier / max-over-ranks / aggregate path of the N-rank
    bench over gloo on CPUs, with a sleep where the engine would run (no compute, no oracle)."""
    from helib_amd import dist as hdist
    if os.environ.get("HX_TEST_FAIL_RANK") == os.environ.get("RANK"):
        # (test hook of the launcher's watchdog: this rank dies during setup, before its first collective)
        print("rank dies during setup (HX_TEST_FAIL_RANK)", file=sys.stderr)
        raise SystemExit(3)
    group = hdist.Group(backend="gloo")
    world, rank = group.world, group.rank
    if args.global_batch:
        start, count = hdist.shard(args.global_batch, world, rank)
    else:
        start, count = rank * args.batch, args.batch
    # the key material of the one key pair: rank 0's words reach every rank (same call as the engine run)
    import numpy as np
    words = np.arange(1, 4097, dtype=np.uint64) * np.uint64(0x9e3779b97f4a7c15) if rank == 0 else None
    words, key_bytes = group.broadcast_words(words, src=0)
    key_sum = int(np.bitwise_xor.reduce(words))
    group.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))
    group.barrier()
    dt = group.max_over_ranks(time.perf_counter() - t0)
    pairs = group.sum_over_ranks(count)
    firsts = group.sum_over_ranks(start if rank == world - 1 else 0)
    want = int(np.bitwise_xor.reduce(np.arange(1, 4097, dtype=np.uint64) * np.uint64(0x9e3779b97f4a7c15)))
    ok_all = int(group.sum_over_ranks(1 if key_sum == want else 0))
    if rank == 0:
        print(json.dumps({"metric": "ctxt_x_ctxt_mults_per_sec_incl_relinearize", "dry_launch": True,
                          "value": round(pairs * args.mults_per_step * args.steps / dt, 1), "unit": "mult/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
                          "scaling": "strong" if args.global_batch else "weak",
                          "config": {"workload": args.workload, "pairs_all_ranks": int(pairs), "last_rank_start": int(firsts),
                                     "batch_this_rank": count, "process_group_world_size": group.world_size_seen(),
                                     "key_material_bytes_broadcast": key_bytes,
                                     "key_material_same_on_all_ranks": bool(ok_all == world)}}))
    group.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0,
                    help="independent ciphertext pairs per GPU per launch set (default 128; ckks65536: 64)")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="split this many pairs over the ranks (strong scaling; configs[3]: 512) instead of --batch per rank")
    ap.add_argument("--mults-per-step", type=int, default=32,
                    help="multiplyBy calls over the resident batch per step (32 x 128 = 4096 multiplications per step)")
    ap.add_argument("--cpu-sample", type=int, default=60, help="multiplies timed on the CPU (0 = skip); 60 = about 15 s of one core")
    ap.add_argument("--ntt-iters", type=int, default=50)
    ap.add_argument("--workload", default="bgv32768", choices=["bgv32768", "bgv32768_fixed", "ckks65536"])
    ap.add_argument("--bits", type=int, default=0,
                    help="ContextBuilder::bits; bgv32768: 950 = the L~16 shape the metric is quoted on, 6400 = the reference's "
                         "own benchmarks/bgv_basic.cpp:247 parameter (L=107, K=36); ckks65536: 1400 (L=24, K=8)")
    ap.add_argument("--precision", type=int, default=1,
                    help="ckks65536: ContextBuilder<CKKS>::precision (1 = the reference's benchmarks/ckks_common.h:45-50)")
    ap.add_argument("--scatter", action="store_true",
                    help="the batch split as a service: rank 0 encrypts ALL pairs and keeps the secret key, the slices travel "
                         "as wire-format ciphertexts (RCCL send / recv), every rank multiplies under PUBLIC key material, "
                         "the products are gathered and rank 0 alone decrypts and verifies them")
    ap.add_argument("--one-device", action="store_true",
                    help="N ranks on GPU 0 (a 1-GPU box): the N-rank path -- one key pair broadcast from rank 0, per-rank "
                         "arenas, barriers -- runs functionally with the real engine; the process group is gloo, because "
                         "RCCL refuses two ranks on one device.  Not a scaling measurement.")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary legs (python mirror, batch-1 latency, op list, levels of the other scheme, config 5)")
    ap.add_argument("--no-rccl-check", action="store_true", help="N = 1: skip the RCCL self-check (config.rccl_selfcheck)")
    ap.add_argument("--dry-launch", action="store_true", help="N-rank launch/aggregation path on CPUs (gloo), no engine")
    ap.add_argument("--cpu-worker", type=int, default=0, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if not args.batch:
        args.batch = 64 if args.workload == "ckks65536" else 128
    if not args.bits:
        args.bits = 1400 if args.workload == "ckks65536" else 950

    if args.cpu_worker:   # one process of the all-cores CPU baseline: no torch, no GPU
        from helib_amd import ctxt as hc
        print(json.dumps(cpu_baseline_fresh(hc.ChainContext(32768, 65537, 1, bits=args.bits, c=3), args.cpu_worker)))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}")
    if args.dry_launch:
        return dry_rank(args)

    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner on fd 1 when a communicator comes
    # up, the HIP runtime and torch now and then a warning -- everything written to fd 1 from here on goes to stderr,
    # the line itself to the saved descriptor
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    from helib_amd import dist as hdist

    world, rank, local_rank = hdist.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    group = (hdist.Group(backend="gloo") if args.one_device else
             hdist.Group(backend="nccl", device=torch.device("cuda", local_rank)))
    from helib_amd import capi as hx, ctxt as hc, host as hh

    B = hdist.shard(args.global_batch, world, rank)[1] if args.global_batch else args.batch
    if B < 1:
        raise SystemExit("bench.py: --global-batch smaller than the number of ranks")
    pairs_all = args.global_batch if args.global_batch else world * args.batch
    R = args.mults_per_step
    rng = np.random.default_rng(1234 + rank)
    sync = torch.cuda.synchronize
    stream = torch.cuda.current_stream().cuda_stream
    extra, roof, cpu = {}, None, None
    extras = not args.no_extras and world == 1            # (N > 1: the ranks end together)
    steps4 = max(1, args.steps // 4)

    if args.workload in ("bgv32768", "ckks65536"):
        ckks = args.workload == "ckks65536"
        # ---- the C++17 host: context, keys, encryptions, the multiply loop, the decryptions ----
        t0 = time.perf_counter()
        sparams = ("ckks", 65536, -1, args.precision, args.bits) if ckks else ("bgv", 32768, 65537, 1, args.bits)
        # ONE key pair for the whole job: rank 0 generates it, its material -- secret key, public
        # encryption key, the relinearisation matrix with its a columns -- is broadcast once (RCCL between device
        # buffers), every other rank builds its session on it and encrypts its own slice of the pairs
        key_bytes = 0
        src, split = None, None
        if args.scatter:
            # the batch split itself (north_star): rank 0 holds every pair and the secret key; what
            # leaves it is PUBLIC key material (broadcast) and each rank's slice of the ciphertexts in the reference's
            # binary format (Ctxt::writeTo), one device tensor per rank and operand
            parts = [hdist.shard(pairs_all, world, r_) for r_ in range(world)]
            assert parts[rank][1] == B
            if rank == 0:
                src = hh.Session(*sparams, pairs_all, device=local_rank, stream=stream, seed=7, source=True)
            keys, key_bytes = group.broadcast_words(src.export_public_keys() if rank == 0 else None, src=0)
            group.barrier()
            ts = time.perf_counter()
            blobs, moved = [], 0
            for which in (0, 1):
                blob, mv = group.scatter_blobs(None, src=0, produce=(lambda r_, w_=which: src.export_ctxts(0, w_, *parts[r_]))
                                               if rank == 0 else None)
                blobs.append(blob)
                moved += mv
            sync()
            group.barrier()
            split = {"scatter_ms": round((time.perf_counter() - ts) * 1e3, 1), "scatter_bytes_this_rank": int(moved),
                     "operand_bytes_this_rank": int(blobs[0].size + blobs[1].size)}
            sess = hh.Session(*sparams, B, device=local_rank, stream=stream, keys=keys, operands=tuple(blobs))
            del blobs, keys
        else:
            if rank == 0:
                sess = hh.Session(*sparams, B, device=local_rank, stream=stream, seed=7)
                keys = sess.export_keys() if world > 1 else None
            if world > 1:
                keys, key_bytes = group.broadcast_words(keys if rank == 0 else None, src=0)
                if rank != 0:
                    sess = hh.Session(*sparams, B, device=local_rank, stream=stream, seed=7 + rank, keys=keys)
                del keys
        sync()
        t_setup = time.perf_counter() - t0

        def verify_level(level):
            """every batch element of the kept product of `level`: each rank its own (it holds the secret key), or --
            --scatter -- the products gathered as wire ciphertexts and rank 0 alone decrypting all of them"""
            if not args.scatter:
                return sess.verify(level), None
            tg = time.perf_counter()
            done = [0]

            def take(r_, blob):
                done[0] += src.verify_blob(blob, level, *parts[r_])
            prod = sess.export_ctxts(level, 0)
            _, mv = group.gather_blobs(prod, dst=0, consume=take if rank == 0 else None)
            return done[0], {"gather_and_verify_ms": round((time.perf_counter() - tg) * 1e3, 1), "product_bytes_this_rank": int(prod.size),
                             "gather_bytes_this_rank": int(mv)}
        n, l, k, d = sess.phim, sess.L_ctxt, sess.K, sess.D
        dtb, host_b = run_session(sess, 1, steps4, args.warmup, R, sync, group.barrier, measure=False)
        dtb = group.max_over_ranks(dtb) / steps4 * args.steps
        host_b = host_b / steps4 * args.steps
        from helib_amd import sensors
        with sensors.Sampler(local_rank) as smp:           # engine clock / package power WHILE the timed loop runs
            dt, host_s = run_session(sess, 1, args.steps, args.warmup, R, sync, group.barrier, measure=True)
        rate_rank = B * R * args.steps / dt                # this rank's own rate over its own clock
        rate_min, rate_max = group.min_over_ranks(rate_rank), group.max_over_ranks(rate_rank)
        dt = group.max_over_ranks(dt)
        nver, gather1 = verify_level(1)                    # every batch element of the last product
        res_primes = sess.result_primes(1)
        prof1 = in_situ_profile(hx, sess, 1, 8, sync) if rank == 0 else None
        # level 2: the kept product with itself (operands that carry the special primes of a key switch)
        run_session(sess, 2, 1, 1, R, sync, group.barrier, measure=True)            # two full warm steps
        malloc0 = sess.arena_stats()["hipMalloc_calls"]
        dt2, _ = run_session(sess, 2, steps4, 0, R, sync, group.barrier, measure=True)
        malloc2 = sess.arena_stats()["hipMalloc_calls"] - malloc0
        dt2 = group.max_over_ranks(dt2)
        nver2, gather2 = verify_level(2)
        prof2 = in_situ_profile(hx, sess, 2, 4, sync, warm=8) if (rank == 0 and extras) else None
        nver_all = int(group.sum_over_ranks(nver))
        nver2_all = int(group.sum_over_ranks(nver2))
        mults = pairs_all * R * args.steps
        scheme = (f"CKKS m=65536 precision={args.precision} bits={args.bits}" if ckks else f"BGV m=32768 p=65537 bits={args.bits}")
        workload = (f"{scheme} (L={l}x{sess.ctxt_bits}b, K={k}x{sess.special_bits}b, {sess.n_small} small primes, D={d}): "
                    + ("Ctxt::multiplyBy on CKKSencrypt-ed ciphertexts = tensorProduct + relin_CKKS_adjust + key switch at the full "
                       "level (a fresh CKKS ciphertext has nothing to mod-switch), benchmarks/ckks_basic.cpp:161-180; "
                       if ckks else
                       "Ctxt::multiplyBy on FRESH ciphertexts = multLowLvl (bringToSet x2 + tensorProduct) + reLinearize "
                       "(dropSmallAndSpecialPrimes + key switch), benchmarks/bgv_basic.cpp:144-165; ")
                    + "added noise MEASURED as in the reference (device canonical-embedding norms, read back lazily); the loop is "
                      "driven by the C++17 host (libhelib_amd_host.so); "
                    f"step = {R} x [copy(ctxt1); copy.multiplyBy(ctxt2)] over the {B}-pair batch -- both operand copies are made "
                    "inside the timed region but are copy-on-write: no bytes move for them; synthetic random plaintexts")
        if ckks:
            # level 1 at the full level: the fixed-level formula; level 2 adds the several-primes mod-switches
            per_mult = algorithmic_bytes_fixed(n, l, k, d)
        else:
            per_mult = algorithmic_bytes_fresh(n, l, k, d)
        extra = {"host": "C++17: include/helib_amd_ctxt.hpp + helib_amd_keys.hpp in helib_amd/lib/libhelib_amd_host.so "
                         "(helib_amd/csrc/host_session.cpp); python only times and checks",
                 "mults_per_step": R, "multiplications_per_step_all_ranks": pairs_all * R,
                 "timed_region_s": round(dt, 3),
                 "bound_noise_mult_per_s": round(mults / dtb, 1),
                 "bound_noise_ms_per_step": round(dtb / args.steps * 1e3, 4),
                 # host time to enqueue a step: with measured noise it contains the waits for the norm read-backs;
                 # the bound-noise run has no waits and is the enqueue cost (throttled by the HIP queue once it is full)
                 "host_ms_per_step_incl_norm_waits": round(host_s / args.steps * 1e3, 4),
                 "host_enqueue_ms_per_step": round(host_b / args.steps * 1e3, 4),
                 "setup_s_keys_and_encryptions": round(t_setup, 2),
                 "one_key_pair": (f"rank 0's key pair for all {world} ranks: {key_bytes} bytes of key material broadcast once "
                                  f"({'gloo, --one-device' if args.one_device else 'RCCL'}), every rank encrypts its own slice under it"
                                  if world > 1 else "single rank"),
                 **smp.summary(),
                 "process_group_world_size": group.world_size_seen(), "world_size_seen": group.world_size_seen(),
                 "key_material_bytes_broadcast": key_bytes,
                 "batch_split": ({"what": "rank 0 encrypts all pairs and keeps the secret key; public key material broadcast; each rank's "
                                          "slice scattered as wire-format ciphertexts (Ctxt::writeTo, one device tensor per rank and "
                                          f"operand, {'gloo' if args.one_device else 'RCCL send/recv'}); products gathered the same way; "
                                          "rank 0 alone decrypts and verifies every product of every rank",
                                  **split, "level1": gather1, "level2": gather2, "verified_on_rank0": True,
                                  "ranks_hold_secret_key": "rank 0 only"} if args.scatter else
                                 "not used: every rank encrypts its own slice (bench.py --scatter moves the ciphertexts instead)"),
                 "per_rank_mult_per_s_min": round(rate_min, 1), "per_rank_mult_per_s_max": round(rate_max, 1),
                 "extras": ("all secondary legs, roofline traffic, cpu_baseline and the RCCL self-check: this is the N = 1 line"
                            if extras else ("skipped at N > 1: the ranks end together, the line carries the headline, the "
                                            "in-situ kernel table and the roofline only" if world > 1 else "skipped (--no-extras)")),
                 "result_primes": res_primes,
                 "inputs": ("key pair, relinearisation matrix and public-key encryptions of random plaintexts made by the C++ "
                            "host (helib_amd_keys.hpp: SecKey::GenSecKey, Encrypt / CKKSencrypt), as benchmarks/bgv_basic.cpp:144-157"),
                 "verified": (f"decrypt(last product) == plaintext product mod (X^N+1{'' if ckks else ', p'}) for all {nver_all} batch "
                              f"elements of all ranks" + (" (decoded: within the error bound the ciphertext reports AND correlated with "
                                                          "the expected product beyond 8 sigma of an unrelated one; smallest correlation "
                                                          f"{getattr(src or sess, 'min_ckks_correlation', float('nan')):.4f})" if ckks else "")
                              + (" -- gathered as wire-format ciphertexts and decrypted by rank 0 alone (the only holder of the secret key)"
                                 if args.scatter else "")),
                 "level2": {"what": "product x product: both operands carry the special primes of the previous key switch "
                                    "(the several-primes mod-switch in front of the tensor product)",
                            "mult_per_s": round(pairs_all * R * steps4 / dt2, 1),
                            "ms_per_mult_of_the_batch": round(dt2 / (steps4 * R) * 1e3, 4),
                            "over_level1": round((dt2 / steps4) / (dt / args.steps), 3),
                            "hipMalloc_calls_in_timed_window": malloc2,
                            "verified_elements": nver2_all, "result_primes": sess.result_primes(2)}}
        if world == 1 and not args.no_rccl_check:
            # RCCL once before the 8-GPU node does it for us: the N-rank run's own calls in a world of one (<= 2 s)
            # (with one real wire-format ciphertext through the scatter / gather code and through an RCCL broadcast)
            extra["rccl_selfcheck"] = hdist.rccl_selfcheck(sess.export_public_keys(), torch.device("cuda", local_rank),
                                                           ctxt_blob=sess.export_ctxts(0, 0, 0, 1))
        extra["level2_mult_per_s"] = extra["level2"]["mult_per_s"]
        extra["level2_over_level1"] = extra["level2"]["over_level1"]
        if rank == 0:
            table, us_per_mult = kernel_table(prof1, n, B, l, k, d, 8)
            extra["kernels_in_situ"] = {"what": "one multiply of the batch, every launch bracketed by HIP events on its stream "
                                                "(8 multiplies profiled after the timed region, behind 24 un-profiled ones)",
                                        "kernel_us_per_multiply_of_the_batch": round(us_per_mult, 1),
                                        "wall_us_per_multiply_of_the_batch": round(dt / (args.steps * R) * 1e6, 1),
                                        "dropped_launch_records": prof1["dropped"], "kernels": table[:16]}
            # the fused sequence's own compulsory bytes: what the kernels that ran are entitled to move (their
            # algorithmic bytes per launch x launches per multiply), per ciphertext multiplication
            fused = sum(r["algorithmic_bytes_per_launch"] * r["launches_per_multiply"] for r in table
                        if "algorithmic_bytes_per_launch" in r) / B
            share = sum(r["share"] for r in table if "algorithmic_bytes_per_launch" in r)
            extra["fused_algorithmic_MB_per_mult"] = round(fused / 1e6, 2)
            extra["fused_hbm_roofline_mult_per_s_per_gpu"] = round(HBM_PEAK_GBS * 1e9 / fused, 0)
            extra["value_over_fused_roofline"] = round((pairs_all / world * R * args.steps / dt) / (HBM_PEAK_GBS * 1e9 / fused), 4)
            extra["fused_bytes_cover_share_of_kernel_time"] = round(share, 3)
            if prof2:
                t2, us2 = kernel_table(prof2, n, B, l, k, d, 4)
                extra["level2"]["kernels_in_situ"] = [{kk: r[kk] for kk in ("kernel", "workgroups", "launches_per_multiply", "avg_us",
                                                                            "us_per_multiply", "share")} for r in t2[:14]]
            b2b = None
            if not ckks:
                # the kernel-level pipeline alone (ctxt primes as rows 0.., specials after) + the back-to-back transform
                cc = hc.ChainContext(32768, 65537, 1, bits=args.bits, c=3)
                shape = dict(M=cc.m, L=l, K=k, digits=[[i - cc.ctxtPrimes[0] for i in dg] for dg in cc.digits])
                sub = hx.Context(cc.m, local_rank)
                fixed_primes = [cc.primes[i] for i in cc.ctxtPrimes + cc.specialPrimes]
                for q in fixed_primes:
                    sub.add_prime(q)
                sub.set_stream(stream)
                if world == 1:
                    dtf = run_fixed(hx, sub, fixed_primes, shape, B, args.steps * R, args.warmup, rng, sync, lambda: None)
                    extra["fixed_level_mult_per_s"] = round(B * R * args.steps / dtf, 1)
                    extra["fixed_level_ms_per_mult_batch"] = round(dtf / (args.steps * R) * 1e3, 4)
                    extra["fixed_level_algorithmic_MB_per_mult"] = round(algorithmic_bytes_fixed(n, l, k, d) / 1e6, 2)
                b2b = ntt_back_to_back(hx, sub, fixed_primes, list(range(l)), list(range(l, l + k)), shape["digits"], B, rng,
                                       args.ntt_iters)
                rec = os.path.join(ROOT, "profiles", "r05_pmc_fresh_multiply_traffic.json")
                if os.path.exists(rec):
                    with open(rec) as f:
                        tr = json.load(f)
                    if tr.get("batch") == B and tr.get("bits") == args.bits:
                        gb = tr["traffic_GB_per_multiply_of_the_batch"]
                        extra["hbm_traffic_GB_per_step_recorded"] = round(gb * R, 1)
                        extra["hbm_traffic_avg_TBps_over_the_step_recorded"] = round(gb * R / (dt / args.steps) / 1e3, 2)
                        extra["hbm_traffic_source"] = ("recorded, not measured in this run: profiles/r05_pmc_fresh_multiply_traffic.json "
                                                       "(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same command, round 5's "
                                                       "kernels -- the multiply path is unchanged since; counted at the L2, so "
                                                       "Infinity-Cache hits are included)")
                        extra["traffic_whole_operation"] = {"GB_per_multiply_of_the_batch": gb, "MB_per_mult": round(gb * 1e3 / B, 2),
                                                            "avg_TBps_over_the_step": extra["hbm_traffic_avg_TBps_over_the_step_recorded"],
                                                            "source": "recorded (profiles/r05_pmc_fresh_multiply_traffic.json), not measured in this run"}
            roof = make_roofline(table, n, B, l, k, d, b2b)
            if extras:
                sync()
                ts = []
                for i in range(14):                      # one ciphertext pair at a time, as benchmarks/bgv_basic.cpp:158-164 times it
                    sync()
                    t0 = time.perf_counter()
                    sess.multiply_single(True)
                    sync()
                    if i >= 2:
                        ts.append((time.perf_counter() - t0) * 1e3)
                ts.sort()
                if not ckks:
                    try:     # batched Encrypt / Decrypt in the C++ host (the reference times one ciphertext per call)
                        extra["batched_encrypt_decrypt"] = sess.encrypt_decrypt_batch(min(B, 64), 3)
                        if not extra["batched_encrypt_decrypt"]["all_elements_round_trip"]:
                            raise SystemExit("bench: DecryptBatch(EncryptBatch(m)) != m")
                    except SystemExit:
                        raise
                    except Exception as e:
                        extra["batched_encrypt_decrypt"] = f"unavailable: {type(e).__name__}: {str(e)[:160]}"
                extra["batch1_latency_ms"] = round(ts[len(ts) // 2], 4)
                extra["batch1_latency_ms_min"] = round(ts[0], 4)
                try:
                    c12 = config1_2_leg(hx)
                    extra.update(c12)
                    c1, c2 = c12["config1_fft_bench"], c12["config2_doublecrt_add_mul"]["batch64"]
                    extra.update({"config1_fft_forward_us_one_row": c1["forward_us_one_row"], "config1_fft_forward_GBps": c1["forward_GBps"],
                                  "config1_fft_inverse_GBps": c1["inverse_GBps"], "config2_add_GBps": c2["add_GBps"],
                                  "config2_mul_GBps": c2["mul_GBps"]})
                except Exception as e:
                    extra["config1_fft_bench"] = f"unavailable: {type(e).__name__}: {str(e)[:160]}"
                try:
                    extra["config5_bluestein"] = config5_leg(hx)
                    extra["config5_bluestein_forward_ns_per_row"] = extra["config5_bluestein"]["forward"]["ns_per_row"]
                    extra["config5_bluestein_inverse_ns_per_row"] = extra["config5_bluestein"]["inverse"]["ns_per_row"]
                except Exception as e:
                    extra["config5_bluestein"] = f"unavailable: {type(e).__name__}: {str(e)[:160]}"
                # the other workloads through the same C++ host, two levels each: the other scheme at its BASELINE shape,
                # and the reference's OWN benchmark parameters (benchmarks/bgv_basic.cpp:247 bits=6400,
                # benchmarks/ckks_basic.cpp:263 precision=1 bits=440)
                legs = {"bgv32768_bits950": (("bgv", 32768, 65537, 1, 950), 128, 8),
                        "ckks65536_bits1400": (("ckks", 65536, -1, args.precision, 1400), 64, 8),
                        "ckks65536_bits440_reference_params": (("ckks", 65536, -1, 1, 440), 64, 8),
                        "bgv32768_bits6400_reference_params": (("bgv", 32768, 65537, 1, 6400), 16, 4),
                        # the reference's general-m benchmark ring (benchmarks/bgv_basic.cpp:236 big_params): m = 32003 prime,
                        # p = 2, bits = 5800 -- every transform a Bluestein convolution
                        "bgv32003_bits5800_reference_params": (("bgv", 32003, 2, 1, 5800), 8, 2),
                        # BASELINE configs[4]'s ring end to end (tests/GTestBootstrapping.cpp:113: m = 21845 = 5 * 17 * 257, p = 2):
                        # every transform of the multiply is the Good-Thomas x Rader kernel (round 6; Bluestein under HX_NO_PFA)
                        "bgv21845_bits950_config5_ring": (("bgv", 21845, 2, 1, 950), 32, 4)}
                mine = "ckks65536_bits%d" % args.bits if ckks else "bgv32768_bits%d" % args.bits
                for name, (sp, bb, rr) in legs.items():
                    if name.startswith(mine):
                        continue
                    try:
                        small = name.startswith("bgv32003")     # (1 warm + 2 timed steps: a multiply takes milliseconds here)
                        leg, so = levels_leg(hh, sp, bb, rr, local_rank, stream, sync, hx=hx,
                                             **(dict(warm_steps=1, steps=2) if small else {}))
                        so.close()
                        del so
                        extra["levels_" + name] = leg
                        for kk in ("level1_mult_per_s", "level2_mult_per_s", "level2_over_level1", "hipMalloc_calls_in_timed_windows"):
                            extra[f"{name}_{kk}"] = leg[kk]          # (flat copies: nested objects do not survive every reader)
                    except Exception as e:
                        extra["levels_" + name] = f"unavailable: {type(e).__name__}: {str(e)[:160]}"
                if ckks:
                    try:
                        extra["ckks_basic_ops"] = ckks_basic_ops(hx, hc, local_rank, stream, sync, args.bits, precision=args.precision)
                    except SystemExit:
                        raise
                    except Exception as e:
                        extra["ckks_basic_ops"] = f"unavailable: {type(e).__name__}: {str(e)[:160]}"
                if not ckks:
                    # the python mirror of the same host logic (helib_amd/ctxt.py + keys.py): secondary figure, the
                    # reference's other benchmark lines, and the HIP-graph replay of one multiply
                    try:
                        cc = hc.ChainContext(32768, 65537, 1, bits=args.bits, c=3)
                        pctx = hx.Context(cc.m, local_rank)
                        for q in cc.primes:
                            pctx.add_prime(q)
                        pctx.set_stream(stream)
                        prepared = real_inputs(hx, hc, cc, pctx, B, 7 + rank)
                        dtp, _, _, nvp, enq = run_fresh(hx, hc, cc, pctx, B, steps4, 1, rng, sync, lambda: None, measure=True,
                                                        prepared=prepared, mults_per_step=R)
                        extra["python_mirror_mult_per_s"] = round(B * R * steps4 / dtp, 1)
                        extra["python_mirror_host_enqueue_ms_per_mult_from_idle"] = round(enq * 1e3, 4)
                        fa, fb, _, sk, msgs = prepared
                        extra["bgv_basic_ops"] = bgv_basic_ops(hx, hc, cc, fa, fb, sk, msgs, sync)
                        _, _, gms = batch1_latency(hx, hc, fa, fb, sync)
                        extra["batch1_latency_hip_graph_ms"] = [round(v, 4) for v in gms] if isinstance(gms, tuple) else gms
                    except SystemExit:
                        raise
                    except Exception as e:
                        extra["python_mirror"] = f"unavailable: {type(e).__name__}: {str(e)[:160]}"
            if args.cpu_sample > 0 and world == 1 and not ckks:
                cc = hc.ChainContext(32768, 65537, 1, bits=args.bits, c=3)
                cpu = cpu_baseline_fresh(cc, max(1, args.cpu_sample // 2))
                cpu["all_cores"] = cpu_baseline_all_cores(args.bits, max(1, args.cpu_sample // 4))
                shape = dict(M=cc.m, L=l, K=k, digits=[[i - cc.ctxtPrimes[0] for i in dg] for dg in cc.digits])
                cpu["fixed_level_value"] = cpu_baseline_fixed(shape, [cc.primes[i] for i in cc.ctxtPrimes + cc.specialPrimes],
                                                              args.cpu_sample)["value"]
            elif args.cpu_sample > 0 and world == 1:
                shape = SHAPES["ckks65536"]
                cpu = cpu_baseline_fixed(shape, gen_primes(shape), max(2, args.cpu_sample // 6))
        sess.close()
    else:
        shape = SHAPES[args.workload]
        primes = gen_primes(shape)
        ctx = hx.Context(shape["M"], local_rank)
        for q in primes:
            ctx.add_prime(q)
        ctx.set_stream(stream)
        n = ctx.phim
        l, k, d = shape["L"], shape["K"], len(shape["digits"])
        dt = run_fixed(hx, ctx, primes, shape, B, args.steps * R, args.warmup, rng, sync, group.barrier)
        dt = group.max_over_ranks(dt)
        workload = shape["name"] + f"; synthetic uniform rows; step = {R} multiplications of the {B}-pair batch"
        per_mult = algorithmic_bytes_fixed(n, l, k, d)
        extra = {"mults_per_step": R, "multiplications_per_step_all_ranks": pairs_all * R,
                 "timed_region_s": round(dt, 3)}
        if rank == 0:
            b2b = ntt_back_to_back(hx, ctx, primes, list(range(l)), list(range(l, l + k)), shape["digits"], B, rng, args.ntt_iters)
            by = b2b["bytes_per_launch"]
            ach = by / (b2b["forward_avg_launch_ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": f"ntt_row_kernel<{n.bit_length() - 1}, false>", "measured": "back to back (HIP events)",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": None, "bytes_per_launch": by, "rows_per_launch": b2b["rows_per_launch"]}
            if args.cpu_sample > 0 and world == 1:
                cpu = cpu_baseline_fixed(shape, primes, args.cpu_sample)

    if rank == 0:
        cfg = {"workload": workload, "batch_per_gpu": B, "pairs_all_ranks": pairs_all,
               "parallelism": (f"replica x{world}, batch-sharded under one key pair (broadcast once), no data-path collective"
                               + (" -- all ranks on ONE device (--one-device): functional run, not a scaling point" if args.one_device else "")),
               "algorithmic_MB_per_mult": round(per_mult / 1e6, 2),
               # (NOT a roofline of this engine: the bytes the REFERENCE's unfused sequence would move, at 8 TB/s -- the fused
               # kernels move fewer, so `value` may exceed it; the engine's own bound is roofline.fused_hbm_roofline_...)
               "unfused_reference_equivalent_hbm_bound_mult_per_s_per_gpu": round(HBM_PEAK_GBS * 1e9 / per_mult, 0),
               "roofline_note": ("algorithmic_MB_per_mult counts the reference-equivalent unfused sequence with "
                                 "per-multiply key rows (SURVEY 8d): a description of the reference's traffic, not a bound on "
                                 "this engine.  roofline.fused_algorithmic_MB_per_mult is the compulsory traffic of the kernels "
                                 "that actually ran (sum of their algorithmic bytes), roofline.value_over_fused_roofline the "
                                 "whole-operation fraction of the 8 TB/s roofline, roofline.whole_operation_traffic the bytes the "
                                 "counters saw; the per-kernel fractions are under `roofline` and `config.kernels_in_situ`")}
        cfg.update(extra)
        if isinstance(roof, dict):
            # the whole operation next to its dominant kernel, at the top level of the line
            for kk in ("fused_algorithmic_MB_per_mult", "fused_hbm_roofline_mult_per_s_per_gpu", "value_over_fused_roofline",
                       "fused_bytes_cover_share_of_kernel_time"):
                if kk in extra:
                    roof[kk] = extra[kk]
            if "traffic_whole_operation" in extra:
                roof["whole_operation_traffic"] = extra["traffic_whole_operation"]
        line = {"metric": "ctxt_x_ctxt_mults_per_sec_incl_relinearize",
                "value": round(pairs_all * R * args.steps / dt, 1), "unit": "mult/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
                "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak",
                "vs_baseline": None, "dtype": "u64",
                "data": "synthetic", "config": cfg, "roofline": roof, "cpu_baseline": cpu}
        line_out.write(json.dumps(line) + "\n")
        line_out.flush()
    group.close()


if __name__ == "__main__":
    main()

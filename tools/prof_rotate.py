"""In-situ kernel profile (hx_profile_begin/_end) of Ctxt::smartAutomorph -- rotate by one generator step -- at the
two benchmark shapes (benchmarks/bgv_basic.cpp / ckks_basic.cpp rotate_a_ciphertext_by1), python mirror of the host.
  python tools/prof_rotate.py [--scheme ckks|bgv] [--batch 16]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scheme", default="ckks")
    ap.add_argument("--batch", type=int, default=16)
    args = ap.parse_args()
    import torch
    from helib_amd import capi as hx, ctxt as hc, keys as hk, hostnt
    ckks = args.scheme == "ckks"
    cc = hc.ChainContext(65536, -1, 20, bits=1400, c=3, ckks=True) if ckks else hc.ChainContext(32768, 65537, 1, bits=950, c=3)
    ctx = hx.Context(cc.m, 0)
    for q in cc.primes:
        ctx.add_prime(q)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    sk = hk.SecKey(cc, hk.HxBackend(ctx, cc), 5)
    sk.GenSecKey(maxDegKswitch=2)
    g = hostnt.ZmStar(cc.m, cc.p).gens[0]
    sk.GenKeySWmatrix(1, g)
    n, B, L = cc.phim, args.batch, len(cc.ctxtPrimes)
    rng = np.random.default_rng(1)
    ct = sk.CKKSencrypt(np.rint(rng.uniform(-1, 1, n) / n * 2**20).astype(np.int64), 1.0, float(2**20)) if ckks else \
        sk.Encrypt(rng.integers(0, 65537, n))
    fa = ct.clone()
    fa.parts = {h: hx.DoubleCRT(ctx, q.getIndexSet(), B, np.repeat(np.asarray(q.download()).reshape(L, -1, n)[:, :1], B, axis=1))
                for h, q in ct.parts.items()}
    fa.ksw_auto[g] = sk.keySwitching[(1, g)].W
    reps = 4
    copies = [fa.clone() for _ in range(reps + 1)]
    copies[0].smartAutomorph(g)
    _ = copies[0].lnNoise
    torch.cuda.synchronize()
    hx.profileBegin()
    t0 = time.perf_counter()
    for c in copies[1:]:
        c.smartAutomorph(g)
    _ = copies[-1].lnNoise
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps * 1e3
    prof = hx.profileEnd()
    ks = [{"kernel": k["kernel"].replace("hx::", "")[:60], "workgroups": k["workgroups"], "calls_per_rotate": k["calls"] / reps,
           "avg_us": round(k["avg_us"], 1), "us_per_rotate": round(k["total_us"] / reps, 1)} for k in prof["kernels"][:16]]
    print(json.dumps({"scheme": args.scheme, "batch": B, "rows": [L, len(cc.specialPrimes), len(cc.digits)], "wall_ms_per_rotate": round(wall, 3),
                      "kernel_ms_per_rotate": round(sum(k["total_us"] for k in prof["kernels"]) / reps / 1e3, 3), "kernels": ks}))


if __name__ == "__main__":
    main()

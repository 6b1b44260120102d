#!/usr/bin/env python3
"""Driver for rocprofv3: HX_ITERS fresh-ciphertext multiplyBy calls (uniform rows, measured
noise) at BGV m=32768 bits=950 over a batch of HX_BATCH pairs -- nothing else, so that a
--kernel-trace or a --pmc pass sees exactly the launch shapes of bench.py's timed loop.
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -- python tools/prof_fresh.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch  # noqa: F401  (loads the HIP runtime first)
    from helib_amd import capi as hx, ctxt as hc
    B = int(os.environ.get("HX_BATCH", "128"))
    iters = int(os.environ.get("HX_ITERS", "2"))
    measure = os.environ.get("HX_MEASURE", "1") != "0"
    cc = hc.ChainContext(32768, 65537, 1, bits=int(os.environ.get("HX_BITS", "950")), c=3)
    ctx = hx.Context(cc.m, 0)
    for q in cc.primes:
        ctx.add_prime(q)
    n = ctx.phim
    rng = np.random.default_rng(1)
    allp = cc.ctxtPrimes + cc.specialPrimes
    D = len(cc.digits)
    kb = np.stack([bench.uniform_rows(rng, cc.primes, allp, 1, n)[:, 0] for _ in range(D)])
    ka = np.stack([bench.uniform_rows(rng, cc.primes, allp, 1, n)[:, 0] for _ in range(D)])
    W = hx.KeySwitch(ctx, allp, kb, ka)
    base = [hx.DoubleCRT(ctx, cc.ctxtPrimes, B, bench.uniform_rows(rng, cc.primes, cc.ctxtPrimes, B, n))
            for _ in range(4)]
    hc.Ctxt.measure = measure
    fa = hc.Ctxt.fresh(cc, hx, base[0], base[1], ksw=W)
    fb = hc.Ctxt.fresh(cc, hx, base[2], base[3], ksw=W)
    for _ in range(iters):
        a = fa.clone()
        a.multiplyBy(fb)
        _ = a.lnNoise
    ctx.sync()
    print("done", B, iters)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Driver for rocprofv3: runs the forward+inverse NTT kernels on the digit-rows shape of one
batched multiply (66 rows x batch) and a few hx_mul_relin steps, nothing else.
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -- python tools/prof_ntt.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch  # noqa: F401  (loads the HIP runtime first)
    from helib_amd import capi as hx
    B = int(os.environ.get("HX_BATCH", "64"))
    iters = int(os.environ.get("HX_ITERS", "5"))
    primes = bench.gen_primes()
    ctx = hx.Context(bench.M, 0)
    for q in primes:
        ctx.add_prime(q)
    n = ctx.phim
    L, K = bench.L, bench.K
    own, sp = list(range(L)), list(range(L, L + K))
    allp = own + sp
    rng = np.random.default_rng(1)
    kb = np.stack([bench.uniform_rows(rng, primes, allp, 1, n)[:, 0] for _ in bench.DIGITS])
    ka = np.stack([bench.uniform_rows(rng, primes, allp, 1, n)[:, 0] for _ in bench.DIGITS])
    W = hx.KeySwitch(ctx, allp, kb, ka)
    polys = [hx.DoubleCRT(ctx, own, B, bench.uniform_rows(rng, primes, own, B, n)) for _ in range(4)]
    o0, o1 = hx.DoubleCRT(ctx, allp, B), hx.DoubleCRT(ctx, allp, B)
    for _ in range(iters):
        hx.multiplyBy(*polys, W, bench.DIGITS, o0, o1)
    ctx.sync()
    print("done", B, iters)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Where does the HOST time of a fresh multiply go?  cProfile over HX_ITERS multiplyBy calls with
noise bounds (no norm waits), batch HX_BATCH; prints the top entries by cumulative time."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch  # noqa: F401
    from helib_amd import capi as hx, ctxt as hc
    B = int(os.environ.get("HX_BATCH", "128"))
    iters = int(os.environ.get("HX_ITERS", "48"))
    cc = hc.ChainContext(32768, 65537, 1, bits=950, c=3)
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
    hc.Ctxt.measure = os.environ.get("HX_MEASURE", "0") != "0"
    fa = hc.Ctxt.fresh(cc, hx, base[0], base[1], ksw=W)
    fb = hc.Ctxt.fresh(cc, hx, base[2], base[3], ksw=W)

    def run(k):
        prev = None
        for _ in range(k):
            a = fa.clone()
            a.multiplyBy(fb)
            prev = a
        return prev

    run(4)
    ctx.sync()
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    run(iters)
    pr.disable()
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"host {1e3 * (t1 - t0) / iters:.3f} ms per multiply, device done after {1e3 * (t2 - t0) / iters:.3f} ms per multiply")
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()

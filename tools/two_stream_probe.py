#!/usr/bin/env python3
"""Probe: do two engine contexts on two HIP streams (each half the batch) fill each other's kernel
tails?  Fresh multiplyBy with noise bounds, uniform rows.  Prints mult/s for 1 context x B and
2 contexts x B/2."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from helib_amd import capi as hx, ctxt as hc
    B = int(os.environ.get("HX_BATCH", "128"))
    K = 16
    hc.Ctxt.measure = False
    cc = hc.ChainContext(32768, 65537, 1, bits=950, c=3)
    rng = np.random.default_rng(1)

    def setup(batch, stream):
        ctx = hx.Context(cc.m, 0)
        for q in cc.primes:
            ctx.add_prime(q)
        ctx.set_stream(stream.cuda_stream)
        n = ctx.phim
        allp = cc.ctxtPrimes + cc.specialPrimes
        D = len(cc.digits)
        kb = np.stack([bench.uniform_rows(rng, cc.primes, allp, 1, n)[:, 0] for _ in range(D)])
        ka = np.stack([bench.uniform_rows(rng, cc.primes, allp, 1, n)[:, 0] for _ in range(D)])
        W = hx.KeySwitch(ctx, allp, kb, ka)
        base = [hx.DoubleCRT(ctx, cc.ctxtPrimes, batch, bench.uniform_rows(rng, cc.primes, cc.ctxtPrimes, batch, n))
                for _ in range(4)]
        fa = hc.Ctxt.fresh(cc, hx, base[0], base[1], ksw=W)
        fb = hc.Ctxt.fresh(cc, hx, base[2], base[3], ksw=W)
        return ctx, fa, fb, W

    def run(sets, k):
        pairs = [[(fa.clone(), fb.clone()) for _ in range(k)] for (_, fa, fb, _) in sets]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            for s in range(len(sets)):
                a, b = pairs[s][i]
                a.multLowLvl(b, destructive=True)
                a.reLinearize()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    s1 = [setup(B, torch.cuda.Stream())]
    run(s1, 3)
    dt1 = run(s1, K)
    print("1 context x", B, ":", round(B * K / dt1, 1), "mult/s")
    del s1
    s2 = [setup(B // 2, torch.cuda.Stream()), setup(B // 2, torch.cuda.Stream())]
    run(s2, 3)
    dt2 = run(s2, K)
    print("2 contexts x", B // 2, ":", round(B * K / dt2, 1), "mult/s")


if __name__ == "__main__":
    main()

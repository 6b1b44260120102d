#!/usr/bin/env python3
"""in-situ launch times of the Good-Thomas x Rader kernels at BASELINE config 5's shape (L = 16, batch 32 = 512 rows)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helib_amd import capi as hx, hostnt
m, L, B = 21845, 16, int(os.environ.get("HX_BATCH", "32"))
g = hostnt.PrimeGen(60, m)
primes = [g.next() for _ in range(L)]
ctx = hx.Context(m)
for p in primes:
    ctx.add_prime(p)
rng = np.random.default_rng(7)
o = np.stack([rng.integers(0, primes[r], size=(B, ctx.phim), dtype=np.uint64) for r in range(L)])
d = hx.DoubleCRT(ctx, list(range(L)), B, o)
for _ in range(10):
    d.FFT(); d.iFFT()
ctx.sync()
hx.profileBegin()
for _ in range(10):
    d.FFT(); d.iFFT()
ctx.sync()
for k in hx.profileEnd()["kernels"]:
    print("%-60s wgs %5d calls %3d avg %7.1f min %7.1f max %7.1f us" % (k["kernel"][:60], k["workgroups"], k["calls"], k["avg_us"], k["min_us"], k["max_us"]))

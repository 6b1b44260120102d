#!/usr/bin/env python3
"""in-situ launch times of the basis extension from n source primes onto nt new ones (addPrimes, src/DoubleCRT.cpp:565-599)
at the shape of the reference's own benchmark chain (benchmarks/bgv_basic.cpp:247, bits = 6400: 36 -> 107, N = 2^14,
batch 16): the matrix-core kernel by default, rns_extend_wide_kernel under HX_NO_MFMA_EXT=1.  No oracle: timing only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helib_amd import capi as hx, hostnt
m = 32768
n, nt, B = int(os.environ.get("HX_N", "36")), int(os.environ.get("HX_NT", "107")), int(os.environ.get("HX_BATCH", "16"))
g = hostnt.PrimeGen(60, m)
primes = [g.next() for _ in range(n + nt)]
ctx = hx.Context(m)
for p in primes:
    ctx.add_prime(p)
rng = np.random.default_rng(7)
src, rest = list(range(n)), list(range(n, n + nt))
o = np.stack([rng.integers(0, primes[r], size=(B, ctx.phim), dtype=np.uint64) for r in src])
reps = 6
ds = [hx.DoubleCRT(ctx, src, B, o) for _ in range(reps + 1)]
ds[0].addPrimes(rest)
ctx.sync()
hx.profileBegin()
for d in ds[1:]:
    d.addPrimes(rest)
ctx.sync()
for k in hx.profileEnd()["kernels"]:
    if "rns_extend" in k["kernel"]:
        print("%-50s wgs %5d calls %3d avg %7.1f min %7.1f max %7.1f us" % (k["kernel"][:50], k["workgroups"], k["calls"], k["avg_us"], k["min_us"], k["max_us"]))

#!/usr/bin/env python3
"""Driver for rocprofv3 / timing of BASELINE configs[4]: Bluestein m=21845 (phi=16384, convolution
length 2^16), DoubleCRT of L=16 primes from PrimeGenerator(60, 21845), batch HX_BATCH (default 32:
512 rows, enough to fill the chip): HX_ITERS forward and inverse transforms back to back.
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace -d gpurun_out/blue_kt -- python tools/prof_bluestein.py
Prints one JSON line with wall-clock timings (algorithmic bytes = 16*N per row)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from helib_amd import capi as hx, hostnt
    m = int(os.environ.get("HX_M", "21845"))
    L = int(os.environ.get("HX_L", "16"))
    B = int(os.environ.get("HX_BATCH", "32"))
    iters = int(os.environ.get("HX_ITERS", "5"))
    rng = np.random.default_rng(7)
    g = hostnt.PrimeGen(60, m)
    primes = [g.next() for _ in range(L)]
    ctx = hx.Context(m)
    for p in primes:
        ctx.add_prime(p)
    n = ctx.phim
    o = np.empty((L, B, n), dtype=np.uint64)
    for r in range(L):
        o[r] = rng.integers(0, primes[r], size=(B, n), dtype=np.uint64)
    d = hx.DoubleCRT(ctx, list(range(L)), B, o)
    d.FFT()
    d.iFFT()
    ctx.sync()
    assert np.array_equal(d.download(), o), "iFFT(FFT(x)) != x"
    # (a stretch of load right before each timed loop: the download + comparison above leave the device idle
    # long enough for its clocks to drop, and the first launches after that are not representative)
    for _ in range(6):
        d.FFT()
        d.iFFT()
    ctx.timerBegin()
    for _ in range(iters):
        d.FFT()
    tf = ctx.timerEnd() / iters * 1e-3
    for _ in range(3):
        d.iFFT()
        d.FFT()
    ctx.timerBegin()
    for _ in range(iters):
        d.iFFT()
    ti = ctx.timerEnd() / iters * 1e-3
    byts = 16 * n * L * B
    print(json.dumps({"config": f"Bluestein m={m} phi={n} L={L} batch={B}", "rows": L * B,
                      "fwd_ms": round(tf * 1e3, 4), "inv_ms": round(ti * 1e3, 4),
                      "fwd_ns_per_row": round(tf / (L * B) * 1e9, 1), "inv_ns_per_row": round(ti / (L * B) * 1e9, 1),
                      "fwd_GBps": round(byts / tf / 1e9, 1), "inv_GBps": round(byts / ti / 1e9, 1),
                      "fwd_frac_of_8TBps": round(byts / tf / 8e12, 4), "inv_frac_of_8TBps": round(byts / ti / 8e12, 4)}))


if __name__ == "__main__":
    main()

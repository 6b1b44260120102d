#!/usr/bin/env python3
"""Timings for the BASELINE.json parity configs that are not the bench.py line:
  config 1  fft_bench   : single-prime NTT, m=16384 (N=8192), 49-bit prime
  config 2  DoubleCRT add / mul, m=32768, L=16
  config 5  Bluestein   : m=21845 DoubleCRT forward+inverse, L=16
Wall-clock over `iters` back-to-back calls between two stream syncs (inputs resident in HBM).
Prints one JSON object per config; GB/s are ALGORITHMIC bytes over time."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timeit(ctx, fn, iters):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / iters


def main():
    from helib_amd import capi as hx, hostnt
    rng = np.random.default_rng(7)
    out = []
    # ---- config 1
    m, B = 16384, 4096
    q = hostnt.PrimeGen(49, m).next()
    ctx = hx.Context(m)
    ctx.add_prime(q)
    n = ctx.phim
    d = hx.DoubleCRT(ctx, [0], B, rng.integers(0, q, size=(1, B, n), dtype=np.uint64))
    tf = timeit(ctx, lambda: d.FFT(), 20)
    ti = timeit(ctx, lambda: d.iFFT(), 20)
    out.append({"config": "fft_bench m=16384 N=8192 49-bit prime", "batch_rows": B,
                "fwd_us_per_row": tf / B * 1e6, "inv_us_per_row": ti / B * 1e6,
                "fwd_GBps": 16 * n * B / tf / 1e9, "inv_GBps": 16 * n * B / ti / 1e9})
    del d, ctx
    # ---- config 2
    m, L, B = 32768, 16, 64
    g = hostnt.PrimeGen(60, m)
    primes = [g.next() for _ in range(L)]
    ctx = hx.Context(m)
    for p in primes:
        ctx.add_prime(p)
    n = ctx.phim
    idx = list(range(L))

    def rows():
        o = np.empty((L, B, n), dtype=np.uint64)
        for r in range(L):
            o[r] = rng.integers(0, primes[r], size=(B, n), dtype=np.uint64)
        return o
    a, b = hx.DoubleCRT(ctx, idx, B, rows()), hx.DoubleCRT(ctx, idx, B, rows())

    def add():
        nonlocal a
        a += b

    def mul():
        nonlocal a
        a *= b
    ta, tm = timeit(ctx, add, 50), timeit(ctx, mul, 50)
    tf = timeit(ctx, lambda: a.FFT(), 20)
    byts = 24 * n * L * B
    out.append({"config": "DoubleCRT add/mul m=32768 L=16", "batch": B,
                "add_us_per_DoubleCRT": ta / B * 1e6, "mul_us_per_DoubleCRT": tm / B * 1e6,
                "add_GBps": byts / ta / 1e9, "mul_GBps": byts / tm / 1e9,
                "ntt_fwd_us_per_DoubleCRT": tf / B * 1e6, "ntt_fwd_GBps": 16 * n * L * B / tf / 1e9})
    del a, b, ctx
    # ---- config 5
    m, L, B = 21845, 16, 8
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
    tf = timeit(ctx, lambda: d.FFT(), 10)
    ti = timeit(ctx, lambda: d.iFFT(), 10)
    out.append({"config": "Bluestein m=21845 phi=16384 L=16 (conv 2^16)", "batch": B,
                "fwd_us_per_DoubleCRT": tf / B * 1e6, "inv_us_per_DoubleCRT": ti / B * 1e6,
                "fwd_GBps": 16 * n * L * B / tf / 1e9, "inv_GBps": 16 * n * L * B / ti / 1e9})
    for x in out:
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in x.items()}))


if __name__ == "__main__":
    main()

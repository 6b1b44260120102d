#!/usr/bin/env python3
"""Timings for SURVEY row N2 at the BASELINE shape (BGV m=32768, p=65537, bits=950): key
generation, PubKey::Encrypt and SecKey::Decrypt (benchmarks/bgv_basic.cpp:186-211 time the same
calls) with the DoubleCRT work on the GPU.  One JSON line per operation.
usage: python tools/bench_keys.py [--reps R]   (device only; the CPU figures quoted in DESIGN.md
section 7.1 came from the same host code over the test oracle's backend)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--bits", type=int, default=950)
    args = ap.parse_args()
    import torch  # noqa: F401
    from helib_amd import capi as hx, ctxt as hc, keys as hk
    cc = hc.ChainContext(32768, 65537, 1, bits=args.bits, c=3)
    gctx = hx.Context(cc.m, 0)
    for q in cc.primes:
        gctx.add_prime(q)
    backends = [("gpu", hk.HxBackend(gctx, cc), gctx.sync)]
    rng = np.random.default_rng(3)
    msg = rng.integers(0, cc.ptxtSpace, size=cc.phim)
    for name, be, sync in backends:
        def timed(fn, reps):
            fn()
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                out = fn()
            sync()
            return (time.perf_counter() - t0) / reps, out
        sk = hk.SecKey(cc, be, seed=1)
        t0 = time.perf_counter()
        sk.GenSecKey(maxDegKswitch=2)
        sync()
        t_keygen = time.perf_counter() - t0
        reps = args.reps if name == "gpu" else 1
        t_enc, ct = timed(lambda: sk.Encrypt(msg), reps)
        t_dec, out = timed(lambda: sk.Decrypt(ct), reps)
        assert out == [int(v) for v in msg]
        a, b = sk.Encrypt(msg), sk.Encrypt(msg)
        a.multiplyBy(b)
        t_dec2, _ = timed(lambda: sk.Decrypt(a), reps)
        rot = [3, 5, 7, 9, 11, 13, 15, 17]
        for k in rot:
            sk.GenKeySWmatrix(1, k)
        cr = sk.Encrypt(msg)

        def plain_rotations():
            outs = []
            for k in rot:
                c = cr.clone()
                c.smartAutomorph(k)
                outs.append(c)
            return outs

        def hoisted_rotations():
            pre = hc.BasicAutomorphPrecon(cr)
            return [pre.automorph(k) for k in rot]
        t_rot, _ = timed(plain_rotations, reps)
        t_hoist, _ = timed(hoisted_rotations, reps)
        for op, t in (("8 rotations, smartAutomorph each (benchmarks/bgv_basic.cpp rotate)", t_rot),
                      ("8 rotations, hoisted (BasicAutomorphPrecon: digits broken once)", t_hoist),
                      ("GenSecKey+pubEncrKey+relin matrix", t_keygen), ("PubKey::Encrypt", t_enc),
                      ("SecKey::Decrypt (fresh, 16 primes)", t_dec),
                      ("SecKey::Decrypt (after multiplyBy, 22 primes)", t_dec2)):
            print(json.dumps({"backend": name, "op": op, "ms": round(t * 1e3, 2),
                              "shape": f"BGV m=32768 bits={args.bits} L={len(cc.ctxtPrimes)} K={len(cc.specialPrimes)}"}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric on MI355X.

metric   : ciphertext x ciphertext multiplications per second, including relinearisation
           (Ctxt::multiplyBy data path: tensorProduct + reLinearize, benchmarks/bgv_basic.cpp:144-165)
workload : BGV m=32768 (N=16384), L=16 x 60-bit ctxt primes, K=6 x 56-bit special primes,
           D=3 digits (6/5/5) -- SURVEY.md Appendix B "bits=950" shape (BASELINE configs[2]).
step     : one hx_mul_relin over a batch of independent ciphertext pairs resident in HBM.
scaling  : weak -- every rank multiplies its own batch; no data-path collective
           (independent ciphertexts shard across GPUs).

Adds "roofline" for the dominant kernel (the forward NTT over the D*(L+K) digit rows, timed
with HIP events on the launch stream) and "cpu_baseline" (the CPU oracle = a port of the
reference algorithm, single thread, bounded sample).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# default workload = BASELINE configs[2] (the configuration the metric is quoted on)
M = 32768
L, K = 16, 6
CT_BITS, SP_BITS = 60, 56
DIGITS = [list(range(0, 6)), list(range(6, 11)), list(range(11, 16))]
WORKLOAD = ("BGV m=32768 N=16384 L=16x60b K=6x56b D=3 (6/5/5) "
            "tensorProduct+reLinearize at fixed level")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def select_workload(name):
    """bgv32768 (default, BASELINE configs[2]) or ckks65536 (configs[3]: m=65536, L=24, K=8,
    D=3 x 8, SURVEY.md Appendix B `bits=1400`; the tensor+relinearise data path is identical,
    CKKS only changes host-side bookkeeping)."""
    global M, L, K, CT_BITS, SP_BITS, DIGITS, WORKLOAD
    if name == "ckks65536":
        M, L, K, CT_BITS, SP_BITS = 65536, 24, 8, 59, 59
        DIGITS = [list(range(0, 8)), list(range(8, 16)), list(range(16, 24))]
        WORKLOAD = ("CKKS m=65536 N=32768 L=24x59b K=8x59b D=3 (8/8/8) "
                    "tensorProduct+reLinearize at fixed level")
    elif name != "bgv32768":
        raise SystemExit(f"unknown workload {name}")


def gen_primes():
    """PrimeGenerator(60, m) x L then PrimeGenerator(56, m) x K, roots by FindPrimRootT.
    Product-side code (helib_amd) supplies its own number theory; the oracle is only
    loaded for the cpu_baseline leg."""
    from helib_amd import hostnt
    g = hostnt.PrimeGen(CT_BITS, M)
    primes = [g.next() for _ in range(L)]
    if SP_BITS == CT_BITS:
        primes += [g.next() for _ in range(K)]
    else:
        g2 = hostnt.PrimeGen(SP_BITS, M)
        primes += [g2.next() for _ in range(K)]
    return primes


def uniform_rows(rng, primes, idx, batch, n):
    out = np.empty((len(idx), batch, n), dtype=np.uint64)
    for r, i in enumerate(idx):
        out[r] = rng.integers(0, primes[i], size=(batch, n), dtype=np.uint64)
    return out


def algorithmic_bytes_per_mult(n, l, k, d):
    """compulsory traffic of one multiply at a fixed level."""
    tensor = l * 56 * n                       # 4 parts in, 3 out (scaling fused)
    ntt = d * (l + k) * 16 * n                # L inverse + D(L+K)-L forward row transforms
    ext = d * (l + k) * 8 * n                 # read each digit's own rows once, write the extension rows
    ks = (l + k) * (3 * d + 4) * 8 * n        # fused inner product incl. own-row rebuild
    return tensor + ntt + ext + ks


def cpu_baseline(primes, sample_mults):
    """Oracle (port of the reference algorithm, -O3, 1 thread) on `sample_mults` multiplies."""
    from oracle import oracle as O
    so_dir = os.path.join(ROOT, "oracle")
    # a native-tuned build of the same C file, made on the machine that runs it
    native = os.path.join(so_dir, "liboracle_native.so")
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=c11", "-shared", "-o",
                               native, os.path.join(so_dir, "hx_oracle.c"), "-lm"],
                              stderr=subprocess.DEVNULL)
        os.environ["HX_ORACLE_SO"] = native
        O._LIB = None
    except Exception:
        pass
    octx = O.Ctx(M)
    for q in primes:
        octx.add_prime(q)
    own, sp = list(range(L)), list(range(L, L + K))
    allp = own + sp
    rng = np.random.default_rng(99)
    n = octx.N
    kb = np.stack([uniform_rows(rng, primes, allp, 1, n)[:, 0] for _ in DIGITS])
    ka = np.stack([uniform_rows(rng, primes, allp, 1, n)[:, 0] for _ in DIGITS])
    ops = [uniform_rows(rng, primes, own, 1, n)[:, 0] for _ in range(4)]
    octx.mul_relin(own, sp, DIGITS, *ops, kb, ka)  # warm
    t0 = time.perf_counter()
    for _ in range(sample_mults):
        octx.mul_relin(own, sp, DIGITS, *ops, kb, ka)
    dt = time.perf_counter() - t0
    return {"value": sample_mults / dt, "unit": "mult/s", "cores": 1, "kind": "port",
            "sample": f"{sample_mults} multiplies (tensor+relinearise) at m={M}, L={L}, K={K}, "
                      f"D={len(DIGITS)}; "
                      "CPU restatement of HElib 2.2.0 algorithms (not NTL), gcc -O3 -march=native, "
                      f"{dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="independent ciphertext pairs per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=8, help="multiplies timed on the CPU (0 = skip)")
    ap.add_argument("--ntt-iters", type=int, default=20)
    ap.add_argument("--workload", default="bgv32768", choices=["bgv32768", "ckks65536"])
    args = ap.parse_args()
    select_workload(args.workload)

    import torch
    from helib_amd import dist as hdist

    world, rank, local_rank = hdist.env_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    torch.cuda.set_device(local_rank)
    group = hdist.Group(backend="nccl", device=torch.device("cuda", local_rank))

    from helib_amd import capi as hx

    primes = gen_primes()
    ctx = hx.Context(M, local_rank)
    for q in primes:
        ctx.add_prime(q)  # root = FindPrimRootT(q, m): the host-supplied root convention
    stream = torch.cuda.current_stream().cuda_stream
    ctx.set_stream(stream)
    n = ctx.phim
    own, sp = list(range(L)), list(range(L, L + K))
    allp = own + sp
    B = args.batch
    rng = np.random.default_rng(1234 + rank)
    kb = np.stack([uniform_rows(rng, primes, allp, 1, n)[:, 0] for _ in DIGITS])
    ka = np.stack([uniform_rows(rng, primes, allp, 1, n)[:, 0] for _ in DIGITS])
    W = hx.KeySwitch(ctx, allp, kb, ka)
    polys = [hx.DoubleCRT(ctx, own, B, uniform_rows(rng, primes, own, B, n)) for _ in range(4)]
    out0 = hx.DoubleCRT(ctx, allp, B)
    out1 = hx.DoubleCRT(ctx, allp, B)

    def step():
        hx.multiplyBy(*polys, W, DIGITS, out0, out1)

    barrier = group.barrier

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    dt = group.max_over_ranks(dt)
    mults = world * B * args.steps
    value = mults / dt

    # ---- roofline of the dominant kernel: forward NTT over the digit rows ----
    roof = None
    cpu = None
    if rank == 0:
        # the forward launch inside one multiply covers the D*(L+K)-L extension rows x B
        nrows = len(DIGITS) * (L + K) - L
        digp = hx.DoubleCRT(ctx, own, B, uniform_rows(rng, primes, own, B, n))
        dg = digp.breakIntoDigits(DIGITS, sp)          # D*(L+K) rows x B, evaluation domain
        hx.time_ntt(dg, True, 2, nrows)                 # warm both directions
        hx.time_ntt(dg, False, 2, nrows)
        ms_inv = hx.time_ntt(dg, True, args.ntt_iters, nrows)
        ms_fwd = hx.time_ntt(dg, False, args.ntt_iters, nrows)
        bytes_launch = 16.0 * n * nrows * B             # 16*N bytes per row transform
        ach = bytes_launch / (ms_fwd * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": f"ntt_row_kernel<{n.bit_length() - 1},fwd>", "achieved": round(ach, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": None, "rows_per_launch": nrows * B,
                "avg_launch_ms": round(ms_fwd, 4), "inverse_avg_launch_ms": round(ms_inv, 4),
                "bytes_per_launch": bytes_launch}
        if args.cpu_sample > 0 and world == 1:
            cpu = cpu_baseline(primes, args.cpu_sample)

    if rank == 0:
        per_mult = algorithmic_bytes_per_mult(n, L, K, len(DIGITS))
        line = {
            "metric": "ctxt_x_ctxt_mults_per_sec_incl_relinearize", "value": round(value, 1),
            "unit": "mult/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": WORKLOAD,
                       "batch_per_gpu": B, "parallelism": f"replica x{world}, batch-sharded",
                       "algorithmic_MB_per_mult": round(per_mult / 1e6, 2),
                       "hbm_roofline_mult_per_s_per_gpu": round(HBM_PEAK_GBS * 1e9 / per_mult, 0)},
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    group.close()


if __name__ == "__main__":
    main()

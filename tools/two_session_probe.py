#!/usr/bin/env python3
"""Probe: two C++ sessions (libhelib_amd_host.so) under ONE key pair on two HIP streams, half the batch each, their
multiplies enqueued alternately -- do the one-round kernels of one (prep, S, norms: ~10 % of a step) hide behind the
other's?  Prints mult/s for 1 x B and 2 x B/2, measured noise and bounds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from helib_amd import host
    B = int(os.environ.get("HX_BATCH", "128"))
    K = int(os.environ.get("HX_K", "256"))
    torch.cuda.set_device(0)
    s0 = torch.cuda.current_stream().cuda_stream

    import threading

    def timed(sessions, measure):
        # one host thread per session (ctypes releases the GIL inside hxh_multiply): a session's wait for its
        # norm read-backs does not keep the other session's kernels from being enqueued
        for s in sessions:
            s.multiply(1, 4, measure)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=lambda s=s: [s.multiply(1, 8, measure) for _ in range(K // 8)]) for s in sessions]
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    one = host.Session("bgv", 32768, 65537, 1, 950, B, stream=s0, seed=7)
    for measure in (True, False):
        dt = timed([one], measure)
        print("1 session x", B, "measure", measure, ":", round(B * K / dt, 1), "mult/s", flush=True)
    keys = one.export_keys()
    one.close()
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    two = [host.Session("bgv", 32768, 65537, 1, 950, B // 2, stream=st[i].cuda_stream, seed=7 + i, keys=(keys if i else None)) for i in range(2)]
    for measure in (True, False):
        dt = timed(two, measure)
        print("2 sessions x", B // 2, "measure", measure, ":", round(B * K / dt, 1), "mult/s", flush=True)
    two2 = None


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Level-1 / level-2 multiplies of the C++ host session with the in-situ kernel table (hx_profile_*).
  python tools/prof_levels.py [bgv|ckks] [batch] [k]
Prints wall time per multiply of the batch for a short and a long loop, the arena's hipMalloc count around
each, and the per-kernel table of one profiled stretch of level-2 multiplies."""
import json
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from helib_amd import capi as hx, host as hh
    scheme = sys.argv[1] if len(sys.argv) > 1 else "ckks"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else (64 if scheme == "ckks" else 128)
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    sync = torch.cuda.synchronize
    stream = torch.cuda.current_stream().cuda_stream
    s = (hh.Session("ckks", 65536, -1, 20, 1400, B, stream=stream, seed=3) if scheme == "ckks" else
         hh.Session("bgv", 32768, 65537, 1, 950, B, stream=stream, seed=3))
    out = {"scheme": scheme, "batch": B}
    for level in (1, 2):
        s.multiply(level, 4, True)
        sync()
        for k in (8, K, K):
            sync()
            t0 = time.perf_counter()
            s.multiply(level, k, True)
            sync()
            out.setdefault(f"level{level}_ms_per_mult", []).append(round((time.perf_counter() - t0) / k * 1e3, 3))
    sync()
    hx.profileBegin()
    s.multiply(2, 4, True)
    sync()
    prof = hx.profileEnd()
    tot = sum(k["total_us"] for k in prof["kernels"])
    out["level2_kernel_us_per_mult"] = round(tot / 4, 1)
    out["level2_kernels"] = [(k["kernel"].replace("hx::", "")[:48], k["workgroups"], k["calls"] // 4, round(k["avg_us"], 1))
                             for k in prof["kernels"][:16]]
    print(json.dumps(out))


if __name__ == "__main__":
    main()

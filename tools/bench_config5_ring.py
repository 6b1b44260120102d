#!/usr/bin/env python3
"""BASELINE configs[4]'s ring end to end: Ctxt::multiplyBy (level 1 and level 2) at m = 21845, p = 2, bits = 950 through the C++
host session bench.py times, batch 32 -- every transform the Good-Thomas x Rader kernel, or Bluestein under HX_NO_PFA=1.
One JSON line (the `levels_bgv21845_bits950_config5_ring` leg of bench.py, alone)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from helib_amd import capi as hx, host as hh
    torch.cuda.set_device(0)
    stream = torch.cuda.current_stream().cuda_stream
    leg, so = bench.levels_leg(hh, ("bgv", 21845, 2, 1, 950), int(os.environ.get("HX_BATCH", "32")), 4, 0, stream,
                               torch.cuda.synchronize, hx=hx)
    so.close()
    leg["transform"] = "Bluestein (HX_NO_PFA)" if os.environ.get("HX_NO_PFA") else "Good-Thomas x Rader"
    print(json.dumps(leg))


if __name__ == "__main__":
    main()

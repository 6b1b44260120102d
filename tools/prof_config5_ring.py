import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from helib_amd import capi as hx, host as hh
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
so = hh.Session("bgv", 21845, 2, 1, 950, 32, device=0, stream=stream, seed=17)
sync = torch.cuda.synchronize
for level in (1, 2):
    so.multiply(level, 4, True); sync()
    prof = bench.in_situ_profile(hx, so, level, 2, sync, warm=4)
    tot = sum(k["total_us"] for k in prof["kernels"])
    print("level", level, "kernel us per multiply of the batch", round(tot / 2, 1))
    for k in prof["kernels"][:22]:
        print("  %-70s wgs %6d calls/mult %5.1f avg %8.1f us  share %.3f" % (k["kernel"][:70], k["workgroups"], k["calls"] / 2, k["avg_us"], k["total_us"] / tot))

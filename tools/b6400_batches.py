import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from helib_amd import capi as hx, host as hh
torch.cuda.set_device(0)
stream = torch.cuda.current_stream().cuda_stream
for B, R in ((16, 4), (48, 2), (64, 2)):
    leg, so = bench.levels_leg(hh, ("bgv", 32768, 65537, 1, 6400), B, R, 0, stream, torch.cuda.synchronize, hx=hx, steps=3)
    so.close(); del so
    print(B, {k: leg[k] for k in leg if 'mult_per_s' in k or 'setup' in k or 'roofline' in k or 'verified' in k})

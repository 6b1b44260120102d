#!/bin/bash
# Does `import torch` complete in a process that has already used libhelib_amd.so?  (a test imports torch late)
# usage: tools/torch_after_lib_probe.sh OUTDIR [HX_LIB=...]
out=gpurun_out/$1; mkdir -p $out; shift
for kv in "$@"; do export "$kv"; done
cat > /tmp/probe.py <<'PY'
import faulthandler, sys, time, os
faulthandler.dump_traceback_later(50, exit=False)
import numpy as np
from helib_amd import capi as hx, hostnt
t0=time.time()
g = hostnt.PrimeGen(60, 16384)
c = hx.Context(16384, 0)
c.add_prime(g.next())
x = np.arange(c.phim, dtype=np.uint64).reshape(1, 1, -1)
d = hx.DoubleCRT(c, [0], 1, x)
d.iFFT(); d.FFT()
print("lib used", round(time.time()-t0,2), flush=True)
t0=time.time()
import torch
print("torch imported", round(time.time()-t0,2), flush=True)
t = torch.zeros(4, device="cuda:0"); torch.cuda.synchronize()
print("torch cuda ok", round(time.time()-t0,2), flush=True)
PY
PYTHONPATH=$PWD python /tmp/probe.py > $out/probe.log 2>&1 &
pid=$!
for i in $(seq 1 40); do sleep 2; kill -0 $pid 2>/dev/null || break; done
if kill -0 $pid 2>/dev/null; then
  echo "STILL RUNNING after 80 s: threads and where they wait" >> $out/probe.log
  for t in /proc/$pid/task/*; do echo "$(basename $t) $(cat $t/comm) wchan=$(cat $t/wchan 2>/dev/null) state=$(grep State $t/status)"; done >> $out/probe.log 2>&1
  which gdb >> $out/probe.log 2>&1 && timeout 60 gdb -batch -ex "thread apply all bt 25" -p $pid >> $out/probe.log 2>&1
  kill -9 $pid
fi
cat $out/probe.log | tail -80

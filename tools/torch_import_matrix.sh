#!/bin/bash
# how long does `import torch` take after various uses of libhelib_amd.so?  (each case limited to LIMIT s)
out=gpurun_out/$1; mkdir -p $out; LIMIT=${2:-70}
export PYTHONPATH=$PWD
run() { # name, python code
  local t0=$(date +%s.%N)
  timeout $LIMIT python -c "$2" > $out/$1.log 2>&1; local rc=$?
  echo "$1 rc=$rc $(echo "$(date +%s.%N) - $t0" | bc) s: $(tail -1 $out/$1.log | cut -c1-150)"
}
T='import time; t0=time.time(); import torch; print("torch imported in", round(time.time()-t0,1))'
run torch_only "$T"
run lib_loaded "from helib_amd import capi; capi.lib(); $T"
run device_count "from helib_amd import capi; print(capi.device_count()); $T"
run context "from helib_amd import capi as hx; c=hx.Context(16384,0); $T"
[ -f helib_amd/lib/variants/unsplit/libhelib_amd.so ] && HX_LIB=$PWD/helib_amd/lib/variants/unsplit/libhelib_amd.so run unsplit_device_count "from helib_amd import capi; print(capi.device_count()); $T"

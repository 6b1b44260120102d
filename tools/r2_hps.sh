#!/bin/bash
# HPS front end of the fast basis-extension kernels: parity suite on the 7-wave build (and the RNS tests
# with the Garner fall-back forced), then level-1 / level-2 multiplies for the committed build (base) and
# the HPS builds at 7 / 6 / 5 waves per SIMD.   gpurun --timeout 600 -- 'bash tools/r2_hps.sh'
export TMPDIR=/tmp
out=gpurun_out/r2hps; mkdir -p $out
V=$PWD/helib_amd/lib/variants
HX_LIB=$V/libhelib_amd_hps7.so timeout 300 python -m pytest tests -m gpu -q -x > $out/pytest_hps7.log 2>&1; echo "pytest hps7 rc=$?"; tail -3 $out/pytest_hps7.log
HX_HPS_EPS=1.0 HX_LIB=$V/libhelib_amd_hps7.so timeout 200 python -m pytest tests -m gpu -q -x -k "scale or digits or bring or switch or poly_rem or add_primes or benchmarked_shape" > $out/pytest_hps7_fallback.log 2>&1; echo "pytest fallback rc=$?"; tail -3 $out/pytest_hps7_fallback.log
for v in base hps7 hps6 hps5; do
  lib=$V/libhelib_amd_$v.so
  b=$(HX_LIB=$lib timeout 150 python tools/bench_levels.py --scheme bgv --m 32768 --bits 950 --batch 128 --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bgv l1', d['level1_ms_per_step'], 'l2', d['level2_ms_per_step'])")
  c=$(HX_LIB=$lib timeout 150 python tools/bench_levels.py --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ckks l1', d['level1_ms_per_step'], 'l2', d['level2_ms_per_step'])")
  echo "$v $b $c" | tee -a $out/hps.log
done

#!/bin/bash
# fast RNS kernels compiled for 8 (default) / 7 / 6 waves per SIMD: fresh multiply and the two level tools
export TMPDIR=/tmp
out=gpurun_out/r2u; mkdir -p $out
for v in default w7 w6 default w7 w6; do
  lib=""; [ $v != default ] && lib=$PWD/helib_amd/lib/variants/libhelib_amd_$v.so
  a=$(HX_LIB=$lib timeout 150 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-extras --inputs uniform 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; print('measured', d['value'], 'fixed', c['fixed_level_mult_per_s'])")
  b=$(HX_LIB=$lib timeout 150 python tools/bench_levels.py --scheme bgv --m 32768 --bits 950 --batch 128 --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bgv l1', d['level1_ms_per_step'], 'l2', d['level2_ms_per_step'])")
  c=$(HX_LIB=$lib timeout 150 python tools/bench_levels.py --steps 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ckks l1', d['level1_ms_per_step'], 'l2', d['level2_ms_per_step'])")
  echo "$v $a $b $c" | tee -a $out/waves.log
done

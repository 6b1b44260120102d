#!/bin/bash
# A/B of kernel build variants on one box: tools/variant_bench.sh name1 name2 ... (names under
# helib_amd/lib/variants/libhelib_amd_<name>.so; "default" = the in-tree build; "env:VAR=1" = the
# in-tree build with that environment variable set).  Two rounds each.  BATCH overrides --batch.
out=gpurun_out/variants.log
mkdir -p gpurun_out; : > $out
for round in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    lib=""; envs=""
    case "$v" in
      default) ;;
      env:*) envs="${v#env:}" ;;
      *) lib=$PWD/helib_amd/lib/variants/libhelib_amd_$v.so ;;
    esac
    env $envs HX_LIB=$lib timeout 150 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-extras --inputs uniform ${BATCH:+--batch $BATCH} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('$v', 'batch', c['batch_per_gpu'], 'measured', d['value'], 'bound', c['bound_noise_mult_per_s'], 'fixed', c['fixed_level_mult_per_s'], 'fwd_ms', r['avg_launch_ms'], 'inv_ms', r['inverse_avg_launch_ms'])" >> $out
  done
done
cat $out

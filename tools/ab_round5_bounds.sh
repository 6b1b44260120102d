#!/bin/bash
# One GPU call of round 5 (kept as the record of how profiles/r05_ab_digit_row_bounds_and_product_on_load.json was
# measured): the fast GPU suite, then same-box runs of the default library against two control builds --
#   tools/build_variant.sh lb8 -DHX_DIGITS_LB8                (digit / extension rows read at bound 8 on every row)
#   tools/build_variant.sh invmulbarrett -DHX_INVMUL_BARRETT  (Barrett product on load in ntt_inv_mul_kernel)
# usage: gpurun -- 'bash tools/ab_round5_bounds.sh'   (writes gpurun_out/r5h/)
bash tools/gpu_calls.sh r5h tests_fast
R=$PWD; out=gpurun_out/r5h
run() { # name workload variant
  v=$3; envs=""
  [ "$v" != default ] && envs="HX_LIB=$R/helib_amd/lib/variants/$v/libhelib_amd.so HX_HOST_LIB=$R/helib_amd/lib/variants/$v/libhelib_amd_host.so"
  f=$out/ab_$1.json
  env $envs timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 ${2:+--workload $2} > $f 2> ${f%.json}.err
  python - $f $1 <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d['config']; l2=c.get('level2') or {}
    ks=[(k['kernel'][:30],k['workgroups'],k['avg_us']) for k in c['kernels_in_situ']['kernels'][:7]]
    print(sys.argv[2],'value',d['value'],'level2',l2.get('mult_per_s'),l2.get('over_level1'),ks)
except Exception as e: print(sys.argv[2],'no line',e)
PY
}
run bgv_default_1 "" default
run bgv_lb8_1 "" lb8
run bgv_default_2 "" default
run bgv_lb8_2 "" lb8
run ckks_default_1 ckks65536 default
run ckks_invmulbarrett_1 ckks65536 invmulbarrett
run ckks_lb8_1 ckks65536 lb8
run ckks_default_2 ckks65536 default

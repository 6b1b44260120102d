# A/B of the N = 2^15 forward transform as two 2^14-point workgroups per row (HX_HALF15=1) vs the one-workgroup kernel
# (default) on config 4 (CKKS m = 65536, bits = 1400, batch 64), same box, alternating.
mkdir -p gpurun_out/ab_half15
for v in half one half2 one2; do
  unset HX_HALF15
  case $v in half*) export HX_HALF15=1;; esac
  timeout 600 python bench.py --workload ckks65536 --steps 6 --warmup 2 --no-extras --cpu-sample 0 --no-rccl-check > gpurun_out/ab_half15/$v.json 2> gpurun_out/ab_half15/$v.err
  python - gpurun_out/ab_half15/$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); c=d['config']
    l2 = c.get('level2') or {}
    print(sys.argv[2],'value',d['value'],'level2',l2.get('mult_per_s'), l2.get('over_level1'))
    for k in (c.get('kernels_in_situ') or {}).get('kernels', [])[:6]: print('   ',k['kernel'][:60],k['workgroups'],k['avg_us'], k.get('frac'))
except Exception as e: print('no line',e)
PY
done

#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3c10
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
for v in off on off on; do
  if [ $v = off ]; then export HX_NO_MULRELIN_FUSE=1; else unset HX_NO_MULRELIN_FUSE; fi
  timeout 300 python bench.py --workload ckks65536 --steps 6 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_ckks_$v.json 2> $out/bench_ckks_$v.err
  python - <<PY
import json
d=json.load(open('$out/bench_ckks_$v.json'))
c=d['config']
ks=[(r['kernel'][:30],r['workgroups'],r['avg_us']) for r in c['kernels_in_situ']['kernels'][:8]]
print('ckks $v', d['value'], c['bound_noise_mult_per_s'], c['level2']['mult_per_s'], c['level2']['over_level1'], ks)
PY
done
unset HX_NO_MULRELIN_FUSE
timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/bench_bgv.json 2> $out/bench_bgv.err
python - <<PY
import json
d=json.load(open('$out/bench_bgv.json'))
c=d['config']
print('bgv', d['value'], c['bound_noise_mult_per_s'], c.get('fixed_level_mult_per_s'), c['level2']['mult_per_s'], d['roofline']['kernel'], d['roofline']['frac'])
PY
HX_NO_MULRELIN_FUSE=1 timeout 300 python bench.py --workload bgv32768_fixed --steps 8 --warmup 3 --cpu-sample 0 > $out/bench_fixed_off.json 2>/dev/null; python -c "import json;print('fixed off', json.load(open('$out/bench_fixed_off.json'))['value'])"
timeout 300 python bench.py --workload bgv32768_fixed --steps 8 --warmup 3 --cpu-sample 0 > $out/bench_fixed_on.json 2>/dev/null; python -c "import json;print('fixed on', json.load(open('$out/bench_fixed_on.json'))['value'])"

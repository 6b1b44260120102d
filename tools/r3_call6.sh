#!/bin/bash
# round 3, GPU call 6: factored-twiddle r16 norm kernels (parity + A/B, BGV and CKKS)
export TMPDIR=/tmp
out=gpurun_out/r3c6
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x -k "norm or ckks or fresh_multiplyBy_at_the_benchmarked or session" > $out/pytest_norm.log 2>&1; echo "pytest norm rc=$?"; tail -4 $out/pytest_norm.log
for v in old new old new; do
  if [ $v = old ]; then export HX_NORM_OLD=1; else unset HX_NORM_OLD; fi
  timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/bench_norm_$v.json 2> $out/bench_norm_$v.err
  python - <<PY
import json
d=json.load(open('$out/bench_norm_$v.json'))
c=d['config']
nk=[(r['kernel'][:40],r['workgroups'],r['avg_us']) for r in c['kernels_in_situ']['kernels'] if 'norm' in r['kernel']]
print('$v', d['value'], c['bound_noise_mult_per_s'], nk)
PY
done
for v in old new; do
  if [ $v = old ]; then export HX_NORM_OLD=1; else unset HX_NORM_OLD; fi
  timeout 300 python bench.py --workload ckks65536 --steps 6 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_ckks_$v.json 2> $out/bench_ckks_$v.err
  python - <<PY
import json
d=json.load(open('$out/bench_ckks_$v.json'))
c=d['config']
nk=[(r['kernel'][:40],r['workgroups'],r['avg_us']) for r in c['kernels_in_situ']['kernels'] if 'norm' in r['kernel']]
print('ckks $v', d['value'], c['bound_noise_mult_per_s'], c['level2']['mult_per_s'], c['level2']['over_level1'], nk)
PY
done

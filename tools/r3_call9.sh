#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3c9
mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
for v in off on off on; do
  if [ $v = off ]; then export HX_NO_LAZY_TENSOR=1; else unset HX_NO_LAZY_TENSOR; fi
  timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/bench_lazy_$v.json 2> $out/bench_lazy_$v.err
  python - <<PY
import json
d=json.load(open('$out/bench_lazy_$v.json'))
c=d['config']
ks=[(r['kernel'][:34],r['workgroups'],r['avg_us']) for r in c['kernels_in_situ']['kernels'][:9]]
print('$v', d['value'], c['bound_noise_mult_per_s'], c['level2']['mult_per_s'], ks)
PY
done

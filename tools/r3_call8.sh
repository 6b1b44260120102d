#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3c8
mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x -k "tensor_folded or cpp_host or session or fresh_multiplyBy or ctxt_multiplyBy or bring_to_set" > $out/pytest_tensor.log 2>&1; echo "pytest tensor rc=$?"; tail -6 $out/pytest_tensor.log
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
unset HX_NO_LAZY_TENSOR
timeout 300 python bench.py --workload ckks65536 --steps 6 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_ckks.json 2> $out/bench_ckks.err
python - <<PY
import json
d=json.load(open('$out/bench_ckks.json'))
c=d['config']
print('ckks', d['value'], c['bound_noise_mult_per_s'], c['level2']['mult_per_s'], c['level2']['over_level1'])
PY

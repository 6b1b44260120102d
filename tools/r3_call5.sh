#!/bin/bash
# round 3, GPU call 5: register-tiled norm kernel (parity + A/B), batched CKKS parity, driver-command trace + PMC passes
export TMPDIR=/tmp
out=gpurun_out/r3c5
mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "norm or ckks_m65536 or fresh_multiplyBy_at_the_benchmarked or bring_to_set_norms or relinearize_norms or break_into_digits" > $out/pytest_norm.log 2>&1; echo "pytest norm rc=$?"; tail -4 $out/pytest_norm.log
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
unset HX_NORM_OLD
(cd /tmp && timeout 500 rocprofv3 --kernel-trace -d $R/$out/kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-sample 0 > $R/$out/bench_traced.json 2> $R/$out/bench_traced.err); echo "trace rc=$?"
python tools/rocpd_summary.py $out/kt --by-grid > $out/bench_kernel_trace.txt 2>&1
python - <<'PY'
import json,re
d=json.load(open('gpurun_out/r3c5/bench_traced.json'))
print('traced value', d['value'], 'roofline', d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
for r in d['config']['kernels_in_situ']['kernels'][:7]:
    print('  in situ', r['kernel'], r['workgroups'], r['avg_us'], r.get('frac'))
PY
grep -E "apply_kernel<14, false>.*wgs|ntt_row_kernel<14, (false|true)>.*wgs +(6400|2048)|keyswitch.*wgs|tensor.*wgs|break_digits.*wgs" $out/bench_kernel_trace.txt
for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  (cd /tmp && timeout 400 rocprofv3 --pmc $ctr -d $R/$out/pmc_$ctr -- python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 --mults-per-step 4 --no-extras --cpu-sample 0 > /dev/null 2> $R/$out/pmc_$ctr.err); echo "pmc $ctr rc=$?"
done
python tools/rocpd_pmc.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_INSTS_VALU > $out/pmc_summary.txt 2>&1; head -60 $out/pmc_summary.txt
find $out -name "*.db" -size +20M -delete

#!/bin/bash
# round 3, GPU call 3: fused Bluestein path (parity, timing, profile), new facade / session tests, profiler agreement
export TMPDIR=/tmp
out=gpurun_out/r3c3
mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "bluestein or general_m or fixture or facade or session or cmodulus or 21845 or norm_general or 1705 or aux" > $out/pytest_blue.log 2>&1; echo "pytest blue rc=$?"; tail -6 $out/pytest_blue.log
timeout 200 python tools/prof_bluestein.py > $out/blue_fused.json 2> $out/blue_fused.err; cat $out/blue_fused.json; tail -3 $out/blue_fused.err
HX_BLUE_OLD=1 timeout 200 python tools/prof_bluestein.py > $out/blue_old.json 2> $out/blue_old.err; cat $out/blue_old.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/$out/blue_kt -- python $R/tools/prof_bluestein.py > /dev/null 2> $R/$out/blue_kt.err); echo "blue kt rc=$?"
python tools/rocpd_summary.py $out/blue_kt > $out/blue_kernel_trace.txt 2>&1; head -24 $out/blue_kernel_trace.txt
(cd /tmp && timeout 500 rocprofv3 --kernel-trace -d $R/$out/kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-sample 0 > $R/$out/bench_traced.json 2> $R/$out/bench_traced.err); echo "trace rc=$?"
python tools/rocpd_summary.py $out/kt --by-grid > $out/bench_kernel_trace.txt 2>&1; grep -E "apply_kernel<14, false>|ntt_row_kernel<14, false>.*6400|keyswitch|tensor" $out/bench_kernel_trace.txt | head
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c3/bench_traced.json'))
print(d['value'])
for r in d['config']['kernels_in_situ']['kernels'][:8]:
    print(r['kernel'], r['workgroups'], r['avg_us'], r.get('frac'))
PY
find $out -name "*.db" -size +20M -delete

#!/bin/bash
# round 3, GPU call 4: side-stream norms A/B, full parity suite, bench lines
export TMPDIR=/tmp
out=gpurun_out/r3c4
mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
HX_NORM_SYNC=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/bench_norm_sync.json 2> $out/bench_norm_sync.err; echo "sync rc=$?"; cut -c1-120 $out/bench_norm_sync.json
timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/bench_norm_async.json 2> $out/bench_norm_async.err; echo "async rc=$?"; cut -c1-120 $out/bench_norm_async.json
HX_NORM_SYNC=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/bench_norm_sync2.json 2> $out/bench_norm_sync2.err; cut -c1-120 $out/bench_norm_sync2.json
timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/bench_norm_async2.json 2> $out/bench_norm_async2.err; cut -c1-120 $out/bench_norm_async2.json
python - <<'PY'
import json
for n in ('sync','async','sync2','async2'):
    d=json.load(open(f'gpurun_out/r3c4/bench_norm_{n}.json'))
    c=d['config']
    print(n, d['value'], c['bound_noise_mult_per_s'], c['level2']['mult_per_s'], d['roofline']['frac'], d['roofline']['avg_launch_us'])
PY
timeout 200 python tools/prof_bluestein.py > $out/blue_fused.json 2> $out/blue_fused.err; cat $out/blue_fused.json
HX_BLUE_OLD=1 timeout 200 python tools/prof_bluestein.py > $out/blue_old.json 2> $out/blue_old.err; cat $out/blue_old.json
timeout 300 python bench.py --workload ckks65536 --steps 6 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_ckks.json 2> $out/bench_ckks.err; echo "ckks rc=$?"; cut -c1-120 $out/bench_ckks.json

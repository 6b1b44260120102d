#!/bin/bash
# Round-2 closing pass, part D: GPU parity suite, smoke(), the default bench line as the driver runs it,
# the fixed-level CKKS workload.   gpurun --timeout 1200 -- 'bash tools/r2_final_d.sh r2z'
export TMPDIR=/tmp
tag=${1:-r2z}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-200 $out/bench.json; tail -2 $out/bench.err
timeout 300 python bench.py --workload ckks65536 --batch 64 --steps 3 --warmup 1 --cpu-sample 0 > $out/bench_ckks_fixed.json 2> $out/bench_ckks_fixed.err; echo "ckks fixed rc=$?"; cut -c1-200 $out/bench_ckks_fixed.json; tail -2 $out/bench_ckks_fixed.err

#!/bin/bash
# GPU parity suite, default bench line, and the bench at batch 256 (no extras) for comparison.
export TMPDIR=/tmp
tag=${1:-r2n}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_gpu.log
timeout 500 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-200 $out/bench.json; tail -3 $out/bench.err
timeout 300 python bench.py --batch 256 --no-extras --cpu-sample 0 --steps 8 > $out/bench_b256.json 2> $out/bench_b256.err; echo "bench256 rc=$?"; cut -c1-200 $out/bench_b256.json
timeout 300 python bench.py --batch 64 --no-extras --cpu-sample 0 --steps 8 > $out/bench_b64.json 2> $out/bench_b64.err; echo "bench64 rc=$?"; cut -c1-200 $out/bench_b64.json

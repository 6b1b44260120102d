#!/bin/bash
export TMPDIR=/tmp
tag=${1:-r2h}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_gpu.log
timeout 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/bench.json; tail -3 $out/bench.err
timeout 200 python tools/bench_levels.py --scheme bgv --m 32768 --bits 950 --batch 128 --steps 6 > $out/bgv.json 2> $out/bgv.err; echo "bgv rc=$?"; cat $out/bgv.json
timeout 200 python tools/bench_levels.py --steps 6 > $out/ckks.json 2> $out/ckks.err; echo "ckks rc=$?"; cat $out/ckks.json

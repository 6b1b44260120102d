#!/bin/bash
# HIP graph tests + the bench line (batch1_latency_hip_graph_ms).   gpurun --timeout 900 -- 'bash tools/r2_graph.sh r2r'
export TMPDIR=/tmp
tag=${1:-r2r}; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q -x -k "graph" > $out/pytest_graph.log 2>&1; echo "pytest rc=$?"; tail -30 $out/pytest_graph.log
timeout 500 python bench.py --steps 6 --warmup 2 --cpu-sample 0 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('$out/bench.json'));c=d['config'];print(d['value'],c.get('batch1_latency_ms'),c.get('batch1_latency_hip_graph_ms'))"; tail -3 $out/bench.err

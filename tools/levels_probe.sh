#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/final4; mkdir -p $out
timeout 150 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['cpu_baseline']['value'], d['cpu_baseline'].get('all_cores'))"; tail -2 $out/bench.err

#!/bin/bash
# GPU parity suite, then level-1 / level-2 multiplies (pipelined) for CKKS configs[3] and BGV configs[1].
export TMPDIR=/tmp
out=gpurun_out/levels; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_gpu.log
timeout 200 python tools/bench_levels.py > $out/ckks3.json 2> $out/ckks3.err; echo "ckks rc=$?"; cat $out/ckks3.json
timeout 200 python tools/bench_levels.py --scheme bgv --m 32768 --bits 950 --batch 128 > $out/bgv3.json 2> $out/bgv3.err; echo "bgv rc=$?"; cat $out/bgv3.json

#!/bin/bash
# GPU parity suite, default bench line, level-1/level-2 timings, and kernel traces of both level tools.
#   gpurun --timeout 1200 -- 'bash tools/r2_pass3.sh r2j'
export TMPDIR=/tmp
tag=${1:-r2j}; out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_gpu.log
timeout 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/bench.json; tail -3 $out/bench.err
timeout 200 python tools/bench_levels.py --scheme bgv --m 32768 --bits 950 --batch 128 --steps 6 > $out/bgv.json 2> $out/bgv.err; echo "bgv rc=$?"; cat $out/bgv.json
timeout 200 python tools/bench_levels.py --steps 6 > $out/ckks.json 2> $out/ckks.err; echo "ckks rc=$?"; cat $out/ckks.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace_bgv -- python $R/tools/bench_levels.py --scheme bgv --m 32768 --bits 950 --batch 128 --steps 3 --warmup 1 > $R/$out/trace_bgv.log 2>&1); echo "trace rc=$?"
python tools/rocpd_summary.py $out/trace_bgv --by-grid > $out/kernel_trace_bgv.txt 2>&1; head -24 $out/kernel_trace_bgv.txt
python tools/level2_sequence.py $out/trace_bgv > $out/level2_sequence_bgv.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace_ckks -- python $R/tools/bench_levels.py --steps 3 --warmup 1 > $R/$out/trace_ckks.log 2>&1); echo "trace rc=$?"
python tools/rocpd_summary.py $out/trace_ckks --by-grid > $out/kernel_trace_ckks.txt 2>&1; head -24 $out/kernel_trace_ckks.txt
python tools/level2_sequence.py $out/trace_ckks > $out/level2_sequence_ckks.txt
find $out -name "*.db" -size +8M -delete

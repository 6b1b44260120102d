#!/bin/bash
# level-1 / level-2 multiplies (BGV m=32768 bits=950 batch 128, CKKS m=65536 bits=1400 batch 64):
# pipelined timings + a kernel trace of each.   gpurun --timeout 700 -- 'bash tools/r2_levels.sh r2e'
export TMPDIR=/tmp
tag=${1:-r2e}; out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 200 python tools/bench_levels.py --scheme bgv --m 32768 --bits 950 --batch 128 --steps 6 > $out/bgv.json 2> $out/bgv.err; echo "bgv rc=$?"; cat $out/bgv.json
timeout 200 python tools/bench_levels.py --steps 6 > $out/ckks.json 2> $out/ckks.err; echo "ckks rc=$?"; cat $out/ckks.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace_bgv -- python $R/tools/bench_levels.py --scheme bgv --m 32768 --bits 950 --batch 128 --steps 3 --warmup 1 > $R/$out/trace_bgv.log 2>&1); echo "trace rc=$?"
python tools/rocpd_summary.py $out/trace_bgv --by-grid > $out/kernel_trace_bgv.txt 2>&1; head -30 $out/kernel_trace_bgv.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace_ckks -- python $R/tools/bench_levels.py --steps 3 --warmup 1 > $R/$out/trace_ckks.log 2>&1); echo "trace rc=$?"
python tools/rocpd_summary.py $out/trace_ckks --by-grid > $out/kernel_trace_ckks.txt 2>&1; head -30 $out/kernel_trace_ckks.txt
find $out -name "*.db" -size +8M -delete

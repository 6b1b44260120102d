#!/bin/bash
export TMPDIR=/tmp
tag=${1:-r2g}; out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace_bgv -- python $R/tools/bench_levels.py --scheme bgv --m 32768 --bits 950 --batch 128 --steps 3 --warmup 1 > $R/$out/trace_bgv.log 2>&1); echo "trace rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace_ckks -- python $R/tools/bench_levels.py --steps 3 --warmup 1 > $R/$out/trace_ckks.log 2>&1); echo "trace rc=$?"
python tools/level2_sequence.py $out/trace_bgv > $out/level2_sequence_bgv.txt; cat $out/level2_sequence_bgv.txt
python tools/level2_sequence.py $out/trace_ckks > $out/level2_sequence_ckks.txt; cat $out/level2_sequence_ckks.txt
find $out -name "*.db" -size +8M -delete

#!/bin/bash
export TMPDIR=/tmp
tag=${1:-r2x}; out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
(cd /tmp && HX_ITERS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace -- python $R/tools/prof_fresh.py > $R/$out/trace.log 2>&1); echo "trace rc=$?"
python tools/rocpd_summary.py $out/trace --by-grid > $out/kernel_trace.txt 2>&1; head -16 $out/kernel_trace.txt
find $out -name "*.db" -size +8M -delete

#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/final3; mkdir -p $out
(cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --cpu-sample 0 > $GRAFT_REPO_ROOT/$out/bench_prof.json 2> $GRAFT_REPO_ROOT/$out/bench_prof.err); echo "prof rc=$?"
python tools/rocpd_summary.py $out/prof --by-grid > $out/kernel_trace.txt 2>&1; tail -16 $out/kernel_trace.txt; cut -c1-120 $out/bench_prof.json

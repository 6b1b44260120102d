#!/bin/bash
# One GPU-box pass: the GPU parity suite, smoke(), the default bench line, the CKKS configs[3]
# sequence, and a rocprofv3 kernel trace of the bench.  Everything lands under gpurun_out/final/.
#   gpurun --timeout 900 -- 'bash tools/final_validation.sh'
export TMPDIR=/tmp
out=gpurun_out/final
mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-300 $out/bench.json
timeout 200 python tools/bench_levels.py --steps 6 > $out/bench_levels_ckks.json 2> $out/bench_levels_ckks.err; echo "ckks rc=$?"; cat $out/bench_levels_ckks.json; tail -3 $out/bench_levels_ckks.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --cpu-sample 0 > $GRAFT_REPO_ROOT/$out/bench_prof.json 2> $GRAFT_REPO_ROOT/$out/bench_prof.err); echo "prof rc=$?"
python tools/rocpd_summary.py $out/prof > $out/kernel_trace.txt 2>&1; head -16 $out/kernel_trace.txt
find $out/prof -name "*.db" -size +20M -delete

#!/bin/bash
# Round-2 closing pass, part C: GPU parity suite, smoke(), the default bench line as the driver runs it,
# and a kernel trace of the fresh multiply.   gpurun --timeout 1200 -- 'bash tools/r2_final_c.sh r2v'
export TMPDIR=/tmp
tag=${1:-r2v}; out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-200 $out/bench.json; tail -2 $out/bench.err
(cd /tmp && HX_ITERS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace -- python $R/tools/prof_fresh.py > $R/$out/trace.log 2>&1); echo "trace rc=$?"
python tools/rocpd_summary.py $out/trace --by-grid > $out/kernel_trace.txt 2>&1; head -16 $out/kernel_trace.txt
find $out -name "*.db" -size +8M -delete

#!/bin/bash
# round 3, GPU call 2: C++ host on the timed path, in-situ profile, arena; GPU parity suite; driver-command trace
export TMPDIR=/tmp
out=gpurun_out/r3c2
mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 400 python bench.py --steps 4 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_quick.json 2> $out/bench_quick.err; echo "bench quick rc=$?"; cut -c1-400 $out/bench_quick.json; tail -5 $out/bench_quick.err
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench_full.json 2> $out/bench_full.err; echo "bench full rc=$?"; cut -c1-300 $out/bench_full.json; tail -5 $out/bench_full.err
timeout 300 python bench.py --workload ckks65536 --steps 6 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_ckks.json 2> $out/bench_ckks.err; echo "bench ckks rc=$?"; cut -c1-300 $out/bench_ckks.json; tail -5 $out/bench_ckks.err
(cd /tmp && timeout 500 rocprofv3 --kernel-trace -d $R/$out/kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --cpu-sample 0 > $R/$out/bench_traced.json 2> $R/$out/bench_traced.err); echo "trace rc=$?"
python tools/rocpd_summary.py $out/kt --by-grid > $out/bench_kernel_trace.txt 2>&1; head -30 $out/bench_kernel_trace.txt
find $out -name "*.db" -size +20M -delete

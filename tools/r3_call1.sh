#!/bin/bash
# round 3, GPU call 1: baselines + the pending norm A/B + Bluestein profile + clocks under load
export TMPDIR=/tmp
out=gpurun_out/r3c1
mkdir -p $out
R=$GRAFT_REPO_ROOT
HX_NORM_SPLIT14=1 timeout 400 python -m pytest tests -m gpu -q -x -k "norm" > $out/pytest_norm_split14.log 2>&1; echo "norm split14 pytest rc=$?"; tail -2 $out/pytest_norm_split14.log
timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_default.json 2> $out/bench_default.err &
BP=$!
sleep 45
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temp|fclk" | head -12; echo ---; sleep 4; done > $out/smi_under_load.txt
wait $BP; echo "bench default rc=$?"; cut -c1-200 $out/bench_default.json
HX_NORM_SPLIT14=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_split14.json 2> $out/bench_split14.err; echo "bench split14 rc=$?"; cut -c1-200 $out/bench_split14.json
timeout 200 python tools/prof_bluestein.py > $out/blue_timing.json 2> $out/blue_timing.err; cat $out/blue_timing.json
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/$out/blue_kt -- python $R/tools/prof_bluestein.py > /dev/null 2> $R/$out/blue_kt.err); echo "blue kt rc=$?"
python tools/rocpd_summary.py $out/blue_kt > $out/blue_kernel_trace.txt 2>&1; head -40 $out/blue_kernel_trace.txt
find $out -name "*.db" -size +20M -delete
(cd tools/ubench && timeout 60 ./bfly_new) > $out/bfly_new.txt 2>&1; cat $out/bfly_new.txt

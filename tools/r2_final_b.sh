#!/bin/bash
# Round-2 closing pass, part B: GPU parity suite, the reference's own bits=6400 parameter, the
# level tool for BGV, a kernel trace and the FETCH_SIZE / WRITE_SIZE passes of the fresh multiply.
#   gpurun --timeout 1500 -- 'bash tools/r2_final_b.sh r2q'
export TMPDIR=/tmp
tag=${1:-r2q}; out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
timeout 500 python bench.py --bits 6400 --batch 16 --steps 5 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_bits6400.json 2> $out/bench_bits6400.err; echo "bits6400 rc=$?"; cut -c1-300 $out/bench_bits6400.json; tail -2 $out/bench_bits6400.err
timeout 200 python tools/bench_levels.py --scheme bgv --m 32768 --bits 950 --batch 128 --steps 6 > $out/bgv.json 2> $out/bgv.err; echo "bgv rc=$?"; cat $out/bgv.json
(cd /tmp && HX_ITERS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace -- python $R/tools/prof_fresh.py > $R/$out/trace.log 2>&1); echo "trace rc=$?"
python tools/rocpd_summary.py $out/trace --by-grid > $out/kernel_trace.txt 2>&1; head -24 $out/kernel_trace.txt
for cnt in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && HX_ITERS=2 timeout 300 rocprofv3 --pmc $cnt -d $R/$out/pmc_$cnt -- python $R/tools/prof_fresh.py > $R/$out/pmc_$cnt.log 2>&1); echo "pmc $cnt rc=$?"
done
python tools/rocpd_pmc.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_hbm_traffic.txt 2>&1; head -70 $out/pmc_hbm_traffic.txt
find $out -name "*.db" -size +8M -delete

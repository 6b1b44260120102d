#!/bin/bash
# One GPU-box pass of round 2: GPU parity suite, default bench line, kernel A/B (round-1 NTT core
# vs new LDS reads vs new LDS reads + multiply-add-chain butterflies), kernel trace and the
# FETCH_SIZE / WRITE_SIZE passes of the fresh multiply.  Everything lands under gpurun_out/$1.
#   gpurun --timeout 1500 -- 'bash tools/r2_gpu_pass.sh r2a'
export TMPDIR=/tmp
tag=${1:-r2a}
out=gpurun_out/$tag
mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
timeout 400 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-1500 $out/bench.json; tail -3 $out/bench.err
if [ -n "$VARIANTS" ]; then
  ROUNDS=${ROUNDS:-1} bash tools/variant_bench.sh $VARIANTS > /dev/null 2>&1; cp gpurun_out/variants.log $out/variants.log; cat $out/variants.log
fi
(cd /tmp && HX_ITERS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace -- python $R/tools/prof_fresh.py > $R/$out/trace.log 2>&1); echo "trace rc=$?"
python tools/rocpd_summary.py $out/trace --by-grid > $out/kernel_trace.txt 2>&1; head -24 $out/kernel_trace.txt
for cnt in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && HX_ITERS=2 timeout 300 rocprofv3 --pmc $cnt -d $R/$out/pmc_$cnt -- python $R/tools/prof_fresh.py > $R/$out/pmc_$cnt.log 2>&1); echo "pmc $cnt rc=$?"
done
python tools/rocpd_pmc.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_hbm_traffic.txt 2>&1; head -60 $out/pmc_hbm_traffic.txt
find $out -name "*.db" -size +8M -delete

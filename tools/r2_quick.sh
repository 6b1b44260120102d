#!/bin/bash
# quick pass: selected GPU tests, kernel A/B, kernel trace of the fresh multiply
export TMPDIR=/tmp
tag=${1:-r2c}; out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x ${KEXPR:+-k "$KEXPR"} > $out/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_sel.log
if [ -n "$VARIANTS" ]; then
  ROUNDS=${ROUNDS:-1} bash tools/variant_bench.sh $VARIANTS > /dev/null 2>&1; cp gpurun_out/variants.log $out/variants.log; cat $out/variants.log
fi
(cd /tmp && HX_ITERS=3 timeout 300 rocprofv3 --kernel-trace --stats -d $R/$out/trace -- python $R/tools/prof_fresh.py > $R/$out/trace.log 2>&1); echo "trace rc=$?"
python tools/rocpd_summary.py $out/trace --by-grid > $out/kernel_trace.txt 2>&1; head -40 $out/kernel_trace.txt
find $out -name "*.db" -size +8M -delete

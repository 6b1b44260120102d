#!/bin/bash
# microbenchmarks + the tests touched since the last pass
export TMPDIR=/tmp
out=gpurun_out/${1:-r2b}; mkdir -p $out
./tools/ubench/issue_bench > $out/issue_bench.txt 2>&1; cat $out/issue_bench.txt
./tools/ubench/bfly_old > $out/bfly.txt 2>&1; ./tools/ubench/bfly_new >> $out/bfly.txt 2>&1; cat $out/bfly.txt
timeout 600 python -m pytest tests -m gpu -q -x -k "randomize or keys or ckks or hoisted or cpp_host" > $out/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest_sel.log

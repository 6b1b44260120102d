#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/levels; mkdir -p $out
timeout 500 python -m pytest tests -m gpu -q -k "cpp_host" > $out/pytest_cpp.log 2>&1; echo "pytest rc=$?"; tail -30 $out/pytest_cpp.log

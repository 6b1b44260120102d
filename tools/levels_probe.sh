#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/levels; mkdir -p $out
timeout 500 python -m pytest tests -m gpu -q -x -k "cpp_host_keys" > $out/pytest_keys_cpp.log 2>&1; echo "pytest rc=$?"; tail -25 $out/pytest_keys_cpp.log

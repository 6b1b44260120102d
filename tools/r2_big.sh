#!/bin/bash
export TMPDIR=/tmp
tag=${1:-r2s}; out=gpurun_out/$tag; mkdir -p $out
timeout 800 python -m pytest tests -m gpu -q -x -k "bluestein or ntt_doublecrt_batched or embedding_norm or general_m" > $out/pytest_big.log 2>&1; echo "pytest rc=$?"; tail -12 $out/pytest_big.log

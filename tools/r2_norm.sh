#!/bin/bash
export TMPDIR=/tmp
tag=${1:-r2w}; out=gpurun_out/$tag; mkdir -p $out
timeout 600 python -m pytest tests -m gpu -q -x -k "norm or benchmarked_shape or ckks_encrypt or cpp_host_ctxt" > $out/pytest_norm.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_norm.log
timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python -c "
import json;d=json.load(open('$out/bench.json'));c=d['config'];print('measured',d['value'],'bounds',c['bound_noise_mult_per_s'],'fixed',c['fixed_level_mult_per_s'],'fwd',d['roofline']['avg_launch_ms'])"
timeout 200 python tools/bench_levels.py --steps 5 > $out/ckks.json 2> $out/ckks.err; python -c "
import json;d=json.load(open('$out/ckks.json'));print('ckks l1',d['level1_ms_per_step'],'l2',d['level2_ms_per_step'])"

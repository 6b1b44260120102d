#!/bin/bash
# Round-2 closing pass, part A: GPU parity suite, smoke(), the default bench line, the same under
# torch.distributed.run (one rank: the RCCL path), the reference's own bits=6400 parameter, configs 1/2/5.
#   gpurun --timeout 1500 -- 'bash tools/r2_final_a.sh r2p'
export TMPDIR=/tmp
tag=${1:-r2p}; out=gpurun_out/$tag; mkdir -p $out
timeout 900 python -m pytest tests -m gpu -q -x > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log
timeout 500 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-200 $out/bench.json; tail -3 $out/bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 4 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_torchrun1.json 2> $out/bench_torchrun1.err; echo "torchrun rc=$?"; cut -c1-200 $out/bench_torchrun1.json; tail -2 $out/bench_torchrun1.err
timeout 400 python bench.py --bits 6400 --batch 16 --steps 5 --warmup 2 --no-extras --cpu-sample 0 > $out/bench_bits6400.json 2> $out/bench_bits6400.err; echo "bits6400 rc=$?"; cut -c1-300 $out/bench_bits6400.json; tail -2 $out/bench_bits6400.err
timeout 300 python tools/bench_configs.py > $out/configs_1_2_5.jsonl 2> $out/configs.err; echo "configs rc=$?"; cat $out/configs_1_2_5.jsonl

#!/bin/bash
# SQ counters of the CKKS level-1 / level-2 multiplies (tools/bench_levels.py at BASELINE configs[3], one
# step per level): instructions per wave and VALU issue rate of the kernels the fresh BGV multiply does not
# run (rns_extend_fast<11>, prep_multi, apply<15,plain>); then the default workload at batch 256.
#   gpurun --timeout 420 -- 'bash tools/r2_pmc_sq_levels.sh r2sq'
export TMPDIR=/tmp
tag=${1:-r2sq}; out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM -d $R/$out/pmc_sq -- python $R/tools/bench_levels.py --steps 1 --warmup 1 > $R/$out/pmc_sq.log 2>&1); echo "pmc sq rc=$?"
python tools/rocpd_pmc.py $out/pmc_sq > $out/pmc_sq_levels_ckks.txt 2>&1; grep -c dispatches $out/pmc_sq_levels_ckks.txt
find $out -name "*.db" -size +8M -delete
for b in 128 256; do
timeout 150 python bench.py --steps 6 --warmup 2 --cpu-sample 0 --no-extras --batch $b 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d['roofline']
print('batch', c['batch_per_gpu'], 'measured', d['value'], 'fixed', c['fixed_level_mult_per_s'], 'fwd_ms', r['avg_launch_ms'], 'frac', r['frac'])" | tee -a $out/batch_ab.log
done

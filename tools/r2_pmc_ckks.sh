#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the N = 2^15 kernels in the CKKS chain of
# BASELINE configs[3] (tools/bench_levels.py, one step per level).
#   gpurun --timeout 600 -- 'bash tools/r2_pmc_ckks.sh r2l'
export TMPDIR=/tmp
tag=${1:-r2l}; out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
for cnt in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 280 rocprofv3 --pmc $cnt -d $R/$out/pmc_$cnt -- python $R/tools/bench_levels.py --steps 1 --warmup 1 > $R/$out/pmc_$cnt.log 2>&1); echo "pmc $cnt rc=$?"
done
python tools/rocpd_pmc.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE > $out/pmc_hbm_traffic_ckks.txt 2>&1; grep -A6 "ntt_row_kernel<15\|apply_kernel<15\|break_digits\|rns_extend_fast_kernel<11\|embed_norm" $out/pmc_hbm_traffic_ckks.txt | head -120
find $out -name "*.db" -size +8M -delete

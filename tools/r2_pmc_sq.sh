#!/bin/bash
# SQ counters of the fresh multiply (tools/prof_fresh.py, batch 128): instruction counts, VALU activity,
# wave residency and waits in one pass; LDS activity / bank conflicts in a second.
#   gpurun --timeout 600 -- 'bash tools/r2_pmc_sq.sh r2y'
export TMPDIR=/tmp
tag=${1:-r2y}; out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
(cd /tmp && HX_ITERS=2 timeout 280 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM -d $R/$out/pmc_sq -- python $R/tools/prof_fresh.py > $R/$out/pmc_sq.log 2>&1); echo "pmc sq rc=$?"
(cd /tmp && HX_ITERS=2 timeout 280 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $R/$out/pmc_lds -- python $R/tools/prof_fresh.py > $R/$out/pmc_lds.log 2>&1); echo "pmc lds rc=$?"
python tools/rocpd_pmc.py $out/pmc_sq $out/pmc_lds > $out/pmc_sq_summary.txt 2>&1; head -90 $out/pmc_sq_summary.txt
tail -3 $out/pmc_lds.log
find $out -name "*.db" -size +8M -delete

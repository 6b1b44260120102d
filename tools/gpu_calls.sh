#!/bin/bash
# The rounds' GPU calls as stages of one script (each gpurun call runs a subset):
#   gpurun --timeout 2400 -- 'bash tools/gpu_calls.sh OUT stage [stage ...]'
# OUT = directory name under gpurun_out/.  Stages:
#   tests            the GPU parity suite (pytest -m gpu), through the C ABI
#   tests:EXPR       the same with -k EXPR
#   tests_fast       the suite without its oracle-heavy cases (the bits = 6400 and m = 32003 chains, whose CPU side takes
#                    minutes: a -k expression that happens to include them cost 8 GPU-minutes once); ~1.5 min
#   smoke            __graft_entry__.smoke()
#   bench            the driver's command, full line  -> bench_full.json
#   bench_ckks       config 4 (--workload ckks65536)   -> bench_ckks.json
#   bench_fixed      the fixed-level multiply          -> bench_fixed.json
#   trace            rocprofv3 --kernel-trace over the driver's command -> bench_kernel_trace.txt + bench_traced.json
#   pmc              three separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU) -> pmc_summary.txt
#   pmc_sq           two more --pmc passes: SQ VALU activity + GRBM_GUI_ACTIVE (clock), LDS bank conflicts -> pmc_sq_summary.txt
#   scatter          bench.py --scatter at N = 1, and at two ranks on one device (gloo)
#   pmc_config5      --pmc passes (traffic, VALU, LDS conflicts) over the config 5 transform driver -> pmc_config5_summary.txt
#   trace_6400       rocprofv3 --kernel-trace over bench.py --bits 6400 --batch 16 (the reference's own bgv_basic parameter) -> bits6400_kernel_trace.txt
#   pmc_mfma         --pmc passes over tools/prof_mfma_ext.py (the matrix-core basis extension, 36 -> 107 primes) -> pmc_mfma_summary.txt
#   bluestein        tools/prof_bluestein.py Good-Thomas x Rader (default), fused Bluestein (HX_NO_PFA) and old chain + kernel trace of the default
#   levels           tools/prof_levels.py for both schemes
#   ab:A,B,...       same-box A/B of the fresh multiply, two rounds; A = default | env:VAR=1 | a variant
#                    directory under helib_amd/lib/variants (tools/build_variant.sh), or variant@VAR=1;
#                    WORKLOAD=ckks65536 for config 4
#   ubench           tools/ubench/bfly_* binaries
#   clocks           rocm-smi engine clock / power samples while the fresh multiply runs -> clocks.txt
export TMPDIR=/tmp
name=$1; shift
out=gpurun_out/$name
mkdir -p $out
R=${GRAFT_REPO_ROOT:-$PWD}
QUICK="--no-extras --cpu-sample 0"

line() {  # one-line digest of a bench JSON
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[2], 'no line:', e); sys.exit(0)
c, r = d['config'], d['roofline']
l2 = c.get('level2') or {}
ks = [(k['kernel'][:34], k['workgroups'], k['avg_us']) for k in (c.get('kernels_in_situ') or {}).get('kernels', [])[:6]]
print(sys.argv[2], 'value', d['value'], 'bound', c.get('bound_noise_mult_per_s'), 'level2', l2.get('mult_per_s'), l2.get('over_level1'),
      'roofline', r.get('kernel'), r.get('avg_launch_us'), r.get('frac'), ks)
PY
}

# canary: some boxes of the pool fault in the runtime's own first copy (every process, whatever it runs) -- give
# such a box back at once instead of spending the call's minutes on it
if ! timeout 120 python - > $out/canary.log 2>&1 <<'PY'
import numpy as np
from helib_amd import capi as hx, hostnt
g = hostnt.PrimeGen(60, 16384)
c = hx.Context(16384, 0)
c.add_prime(g.next())
x = np.arange(c.phim, dtype=np.uint64).reshape(1, 1, -1)
d = hx.DoubleCRT(c, [0], 1, x)
d.iFFT(); d.FFT()
assert np.array_equal(d.download(), x)
print("canary ok")
PY
then
  echo "CANARY FAILED: this box cannot run a single transform -- giving it back"; tail -3 $out/canary.log; exit 9
fi

for st in "$@"; do
  case "$st" in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest_gpu.log ;;
    tests_fast)
      # (--timeout: a test that hangs costs its own limit, with the stack of every thread in the log, not the whole call's)
      timeout 900 python -m pytest tests -m gpu -q -x --timeout 150 -k "not reference_benchmark_chain_size and not 32003 and not 5800 and not 6400" > $out/pytest_fast.log 2>&1; echo "pytest fast rc=$?"; tail -3 $out/pytest_fast.log ;;
    tests:*)
      timeout 1200 python -m pytest tests -m gpu -q -x --timeout 300 -k "${st#tests:}" > $out/pytest_k.log 2>&1; echo "pytest -k rc=$?"; tail -4 $out/pytest_k.log ;;
    smoke)
      timeout 300 python __graft_entry__.py --smoke > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/smoke.log ;;
    bench)
      timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_full.json 2> $out/bench_full.err; echo "bench rc=$?"
      line $out/bench_full.json bgv; tail -3 $out/bench_full.err ;;
    bench_ckks)
      timeout 600 python bench.py --workload ckks65536 --steps 8 --warmup 3 > $out/bench_ckks.json 2> $out/bench_ckks.err; echo "bench ckks rc=$?"
      line $out/bench_ckks.json ckks; tail -3 $out/bench_ckks.err ;;
    ranks2)
      # the N-rank path with the real engine on this 1-GPU box: two ranks on GPU 0, one key pair broadcast from rank 0
      timeout 600 python bench.py --gpus 2 --one-device --batch 16 --steps 3 --warmup 1 --mults-per-step 8 --no-extras --cpu-sample 0 \
         > $out/bench_ranks2.json 2> $out/bench_ranks2.err; echo "ranks2 rc=$?"
      python - $out/bench_ranks2.json <<'PY'
import json, sys
try:
    d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith('{')][-1]); c = d['config']
    print('ranks2: value', d['value'], 'n_gpus', d['n_gpus'], 'world', c.get('process_group_world_size'), 'key bytes', c.get('key_material_bytes_broadcast'), c.get('verified'))
except Exception as e:
    print('ranks2: no line', e)
PY
      tail -3 $out/bench_ranks2.err ;;
    scatter)
      # the batch split as a service on this 1-GPU box: (a) one rank, source + worker session on the device;
      # (b) two ranks on GPU 0 over gloo: slices scattered, products gathered, rank 0 alone verifies
      timeout 600 python bench.py --scatter --steps 4 --warmup 1 --mults-per-step 8 --no-extras --cpu-sample 0 \
         > $out/bench_scatter1.json 2> $out/bench_scatter1.err; echo "scatter1 rc=$?"
      timeout 900 python bench.py --gpus 2 --one-device --scatter --workload ckks65536 --batch 16 --steps 3 --warmup 1 --mults-per-step 8 --no-extras --cpu-sample 0 \
         > $out/bench_scatter2.json 2> $out/bench_scatter2.err; echo "scatter2 rc=$?"
      python - $out/bench_scatter1.json $out/bench_scatter2.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([ln for ln in open(f) if ln.startswith('{')][-1]); c = d['config']
        print(f.split('/')[-1], 'value', d['value'], 'n_gpus', d['n_gpus'], 'split', json.dumps(c.get('batch_split'))[:600], c.get('verified'))
    except Exception as e:
        print(f, 'no line', e)
PY
      tail -n 3 $out/bench_scatter1.err; tail -n 3 $out/bench_scatter2.err ;;
    bench6400)
      # the reference's own BGV parameter (benchmarks/bgv_basic.cpp:247): bits=6400 -> L=107, K=36, D=3
      timeout 900 python bench.py --bits 6400 --batch 16 --steps 4 --warmup 1 --mults-per-step 4 --no-extras --cpu-sample 0 \
         > $out/bench_6400.json 2> $out/bench_6400.err; echo "bench6400 rc=$?"
      line $out/bench_6400.json bits6400; tail -3 $out/bench_6400.err ;;
    trace_ckks)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$out/kt_ckks -- python3 $R/bench.py --workload ckks65536 --steps 8 --warmup 3 $QUICK \
         > $R/$out/bench_ckks_traced.json 2> $R/$out/bench_ckks_traced.err); echo "trace_ckks rc=$?"
      python tools/rocpd_summary.py $out/kt_ckks --by-grid > $out/bench_ckks_kernel_trace.txt 2>&1
      grep -E "wgs" $out/bench_ckks_kernel_trace.txt | head -14 ;;
    trace6400)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$out/kt_6400 -- python3 $R/bench.py --bits 6400 --batch 16 --steps 4 --warmup 1 --mults-per-step 4 $QUICK \
         > $R/$out/bench_6400_traced.json 2> $R/$out/bench_6400_traced.err); echo "trace6400 rc=$?"
      python tools/rocpd_summary.py $out/kt_6400 --by-grid > $out/bench_6400_kernel_trace.txt 2>&1
      grep -E "wgs" $out/bench_6400_kernel_trace.txt | head -14 ;;
    bench_fixed)
      timeout 400 python bench.py --workload bgv32768_fixed --steps 8 --warmup 3 --cpu-sample 0 > $out/bench_fixed.json 2> $out/bench_fixed.err
      python -c "import json;print('fixed level', json.load(open('$out/bench_fixed.json'))['value'])" ;;
    trace)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$out/kt -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 $QUICK \
         > $R/$out/bench_traced.json 2> $R/$out/bench_traced.err); echo "trace rc=$?"
      python tools/rocpd_summary.py $out/kt --by-grid > $out/bench_kernel_trace.txt 2>&1
      line $out/bench_traced.json traced
      grep -E "wgs" $out/bench_kernel_trace.txt | head -16 ;;
    trace_6400)
      (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$out/kt6400 -- python3 $R/bench.py --gpus 1 --bits 6400 --batch 16 --steps 6 --warmup 2 \
         --mults-per-step 4 $QUICK --no-rccl-check > $R/$out/bench_bits6400_traced.json 2> $R/$out/bench_bits6400_traced.err); echo "trace_6400 rc=$?"
      python tools/rocpd_summary.py $out/kt6400 --by-grid > $out/bits6400_kernel_trace.txt 2>&1
      line $out/bench_bits6400_traced.json traced_6400
      grep -E "wgs" $out/bits6400_kernel_trace.txt | grep -E "rns_extend|apply|keyswitch|ntt_row" | head -16 ;;
    pmc)
      for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
        (cd /tmp && timeout 400 rocprofv3 --pmc $ctr -d $R/$out/pmc_$ctr -- python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 \
           --mults-per-step 4 $QUICK > /dev/null 2> $R/$out/pmc_$ctr.err); echo "pmc $ctr rc=$?"
      done
      python tools/rocpd_pmc.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_INSTS_VALU > $out/pmc_summary.txt 2>&1
      head -40 $out/pmc_summary.txt ;;
    pmc_sq)
      # two more passes: VALU activity and the clock the kernels actually ran at; LDS conflicts (norm kernels)
      i=0
      for ctrs in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS"; do
        i=$((i+1))
        (cd /tmp && timeout 400 rocprofv3 --pmc $ctrs -d $R/$out/pmc_sq$i -- python3 $R/bench.py --gpus 1 --steps 2 --warmup 1 \
           --mults-per-step 4 $QUICK > /dev/null 2> $R/$out/pmc_sq$i.err); echo "pmc_sq pass $i rc=$?"
      done
      python tools/rocpd_pmc.py $out/pmc_sq1 $out/pmc_sq2 > $out/pmc_sq_summary.txt 2>&1
      grep -E "ntt_row_kernel<14, false>.* 6400 |apply_kernel<14, false>|embed_norm" $out/pmc_sq_summary.txt | cut -c1-200 | head -40 ;;
    bluestein)
      timeout 200 python tools/prof_bluestein.py > $out/blue_pfa.json 2> $out/blue_pfa.err; cat $out/blue_pfa.json
      HX_PFA_NO_REM=1 timeout 200 python tools/prof_bluestein.py > $out/blue_pfa_conv_rem.json 2> $out/blue_pfa_conv_rem.err; cat $out/blue_pfa_conv_rem.json
      HX_NO_PFA=1 timeout 200 python tools/prof_bluestein.py > $out/blue_fused.json 2> $out/blue_fused.err; cat $out/blue_fused.json
      HX_BLUE_OLD=1 timeout 200 python tools/prof_bluestein.py > $out/blue_old.json 2> $out/blue_old.err; cat $out/blue_old.json
      (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $R/$out/blue_kt -- python3 $R/tools/prof_bluestein.py > /dev/null 2> $R/$out/blue_kt.err)
      python tools/rocpd_summary.py $out/blue_kt > $out/blue_kernel_trace.txt 2>&1; head -14 $out/blue_kernel_trace.txt ;;
    pmc_config5)
      # separate --pmc passes over the config 5 transform driver (tools/prof_bluestein.py): traffic, VALU instructions, LDS
      for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
        tag=$(echo $ctr | cut -d" " -f1)
        (cd /tmp && HX_ITERS=3 timeout 300 rocprofv3 --pmc $ctr -d $R/$out/pmc5_$tag -- python3 $R/tools/prof_bluestein.py > /dev/null 2> $R/$out/pmc5_$tag.err); echo "pmc5 $tag rc=$?"
      done
      python tools/rocpd_pmc.py $out/pmc5_FETCH_SIZE $out/pmc5_WRITE_SIZE $out/pmc5_SQ_INSTS_VALU $out/pmc5_SQ_LDS_BANK_CONFLICT > $out/pmc_config5_summary.txt 2>&1
      grep -E "pfa_row" $out/pmc_config5_summary.txt | cut -c1-220 ;;
    pmc_mfma)
      for ctr in "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" FETCH_SIZE WRITE_SIZE; do
        tag=$(echo $ctr | cut -d" " -f1)
        (cd /tmp && timeout 300 rocprofv3 --pmc $ctr -d $R/$out/pmcm_$tag -- python3 $R/tools/prof_mfma_ext.py > /dev/null 2> $R/$out/pmcm_$tag.err); echo "pmc_mfma $tag rc=$?"
      done
      python tools/rocpd_pmc.py $out/pmcm_SQ_INSTS_VALU $out/pmcm_SQ_ACTIVE_INST_VALU $out/pmcm_SQ_VALU_MFMA_BUSY_CYCLES $out/pmcm_FETCH_SIZE $out/pmcm_WRITE_SIZE > $out/pmc_mfma_summary.txt 2>&1
      grep -E "rns_extend_mfma|rns_extend_wide" $out/pmc_mfma_summary.txt | cut -c1-220 ;;
    levels)
      timeout 300 python tools/prof_levels.py bgv > $out/levels_bgv.json 2> $out/levels_bgv.err; cut -c1-1500 $out/levels_bgv.json
      timeout 300 python tools/prof_levels.py ckks > $out/levels_ckks.json 2> $out/levels_ckks.err; cut -c1-1500 $out/levels_ckks.json ;;
    ab:*)
      IFS=, read -ra VS <<< "${st#ab:}"
      for round in 1 2; do
        for v in "${VS[@]}"; do
          envs=""
          case "$v" in
            default) ;;
            env:*) envs="${v#env:}" ;;
            *@*) envs="HX_LIB=$R/helib_amd/lib/variants/${v%%@*}/libhelib_amd.so HX_HOST_LIB=$R/helib_amd/lib/variants/${v%%@*}/libhelib_amd_host.so ${v#*@}" ;;
            *) envs="HX_LIB=$R/helib_amd/lib/variants/$v/libhelib_amd.so HX_HOST_LIB=$R/helib_amd/lib/variants/$v/libhelib_amd_host.so" ;;
          esac
          f=$out/ab_${v//[^A-Za-z0-9_]/_}_$round.json
          env $envs timeout 300 python bench.py --steps 8 --warmup 3 $QUICK ${WORKLOAD:+--workload $WORKLOAD} > $f 2> ${f%.json}.err
          line $f "$v#$round"
        done
      done ;;
    clocks)
      (for i in $(seq 1 40); do rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo; sleep 0.5; done) > $out/clocks.txt &
      smi=$!
      timeout 300 python bench.py --steps 12 --warmup 3 $QUICK > $out/bench_clocks.json 2> $out/bench_clocks.err
      wait $smi
      line $out/bench_clocks.json clocks; sed -n '1p;10p;20p;30p;40p' $out/clocks.txt ;;
    ubench)
      for b in tools/ubench/bfly_*; do [ -x $b ] && { echo "== $b"; timeout 60 $b; }; done > $out/ubench.txt 2>&1; cat $out/ubench.txt ;;
    *) echo "unknown stage $st" ;;
  esac
done
find $out -name "*.db" -size +20M -delete

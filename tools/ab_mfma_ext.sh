# A/B of the basis extension from 36 source primes, same box, bits = 6400 at batch 16: the leg's rates and the in-situ
# kernel table.  Variants: mfma (default), valu (HX_NO_MFMA_EXT=1: rns_extend_wide_kernel), and any variant library
# named on the command line (helib_amd/lib/variants/NAME, tools/mfma_variant.sh).  Output under gpurun_out/ab_mfma/.
mkdir -p gpurun_out/ab_mfma
for v in mfma valu "$@" mfma_again; do
  unset HX_NO_MFMA_EXT HX_LIB HX_HOST_LIB
  case $v in
    valu*) export HX_NO_MFMA_EXT=1;;
    mfma*) ;;
    *) export HX_LIB=$PWD/helib_amd/lib/variants/$v/libhelib_amd.so HX_HOST_LIB=$PWD/helib_amd/lib/variants/$v/libhelib_amd_host.so;;
  esac
  timeout 600 python bench.py --bits 6400 --batch 16 --steps 4 --warmup 1 --mults-per-step 4 --no-extras --cpu-sample 0 --no-rccl-check > gpurun_out/ab_mfma/$v.json 2> gpurun_out/ab_mfma/$v.err
  python - gpurun_out/ab_mfma/$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); c=d['config']
    print(sys.argv[2],'value',d['value'],'level2',c.get('level2_mult_per_s'))
    for k in c['kernels_in_situ']['kernels'][:12]:
        if 'rns_extend' in k['kernel']: print('   ',k['kernel'][:60],k['workgroups'],k['launches_per_multiply'],k['avg_us'],k['min_us'])
except Exception as e: print('no line',e)
PY
done

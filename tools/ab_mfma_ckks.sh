# A/B of the matrix-core basis extension on the CKKS chain (config 4: m = 65536, bits = 1400, batch 64), whose level-2
# mod-switch drops 8 + 3 primes: default vs HX_NO_MFMA_EXT=1 (rns_extend_fast_kernel<11, HPS>), same box, alternating.
mkdir -p gpurun_out/ab_mfma_ckks
for v in mfma valu mfma2 valu2; do
  unset HX_NO_MFMA_EXT
  case $v in valu*) export HX_NO_MFMA_EXT=1;; esac
  timeout 600 python bench.py --workload ckks65536 --steps 6 --warmup 2 --no-extras --cpu-sample 0 --no-rccl-check > gpurun_out/ab_mfma_ckks/$v.json 2> gpurun_out/ab_mfma_ckks/$v.err
  python - gpurun_out/ab_mfma_ckks/$v.json $v <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); c=d['config']
    l2 = c.get('level2') or {}
    print(sys.argv[2],'value',d['value'],'level2',l2.get('mult_per_s'), l2.get('over_level1'))
    for k in (l2.get('kernels_in_situ') or c.get('kernels_in_situ') or {}).get('kernels', [])[:12]:
        if 'rns_extend' in k['kernel']: print('   ',k['kernel'][:60],k['workgroups'],k['launches_per_multiply'],k['avg_us'])
except Exception as e: print('no line',e)
PY
done

mkdir -p gpurun_out/r5_6400
for b in 16 48 96; do
  timeout 600 python bench.py --bits 6400 --batch $b --steps 4 --warmup 1 --mults-per-step 4 --no-extras --cpu-sample 0 --no-rccl-check > gpurun_out/r5_6400/b$b.json 2> gpurun_out/r5_6400/b$b.err
  python - gpurun_out/r5_6400/b$b.json $b <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline()); c=d['config']
    print('batch',sys.argv[2],'value',d['value'],'level2',c.get('level2_mult_per_s'),[(k['kernel'][:30],k['workgroups'],k['avg_us']) for k in c['kernels_in_situ']['kernels'][:6]])
except Exception as e: print('no line',e)
PY
done

export TMPDIR=/tmp
out=gpurun_out/r06_hpsmin; mkdir -p $out
for round in 1 2; do
for v in 9 8 5; do
  HX_HPS_MIN_N=$v timeout 300 python bench.py --workload ckks65536 --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/ckks_h${v}_$round.json 2> $out/ckks_h${v}_$round.err
  HX_HPS_MIN_N=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/bgv_h${v}_$round.json 2> $out/bgv_h${v}_$round.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_hpsmin/*.json')):
    try:
        d=json.load(open(f)); c=d['config']
        ks=[(k['kernel'][:34],k['workgroups'],k['avg_us']) for k in c['kernels_in_situ']['kernels'] if 'break' in k['kernel']]
        print(f.split('/')[-1], d['value'], c['level2']['mult_per_s'], ks)
    except Exception as e: print(f,'ERR',e)
PY

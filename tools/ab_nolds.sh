export TMPDIR=/tmp
out=gpurun_out/r06_nolds; mkdir -p $out
HX_BRK_NOLDS=1 timeout 900 python -m pytest tests -m gpu -q -x --timeout 300 -k "break_into_digits or multiply_relin or ckks_m65536 or timed_launch_shapes or tensor_and_keyswitch or hps_form" > $out/pytest_nolds.log 2>&1; echo "pytest nolds rc=$?"; tail -3 $out/pytest_nolds.log
for round in 1 2; do
for v in 0 1; do
  HX_BRK_NOLDS=$v timeout 300 python bench.py --workload ckks65536 --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/ckks_nolds${v}_$round.json 2> $out/ckks_nolds${v}_$round.err
  HX_BRK_NOLDS=$v timeout 300 python bench.py --steps 8 --warmup 3 --no-extras --cpu-sample 0 > $out/bgv_nolds${v}_$round.json 2> $out/bgv_nolds${v}_$round.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06_nolds/*_nolds*_*.json')):
    try:
        d=json.load(open(f)); c=d['config']
        ks=[(k['kernel'][:30],k['workgroups'],k['avg_us']) for k in c['kernels_in_situ']['kernels'] if 'break' in k['kernel']]
        print(f.split('/')[-1], d['value'], c['level2']['mult_per_s'], c['level2']['over_level1'], ks)
    except Exception as e: print(f,'ERR',e)
PY

#!/bin/bash
# occupancy experiment: the plain row transforms at 2 workgroups per CU (default) and at 1 (padded LDS)
export TMPDIR=/tmp
tag=${1:-r2t}; out=gpurun_out/$tag; mkdir -p $out
for pad in 0 40000; do
  HX_NTT_LDS_PAD=$pad timeout 200 python bench.py --workload bgv32768_fixed --no-extras --cpu-sample 0 --steps 3 --warmup 1 > $out/fixed_pad$pad.json 2> $out/fixed_pad$pad.err
  python -c "
import json;d=json.load(open('$out/fixed_pad$pad.json'));r=d['roofline'];print('pad',$pad,'fwd ms',r['avg_launch_ms'],'inv ms',r['inverse_avg_launch_ms'],'value',d['value'])"
done

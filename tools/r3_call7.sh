#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/r3c7
mkdir -p $out
HX_ARENA_TRACE=1 HX_NORM_OLD=1 timeout 300 python tools/prof_levels.py ckks 64 32 > $out/levels_ckks_old.json 2> $out/levels_ckks_old.err; cat $out/levels_ckks_old.json; grep -c arena $out/levels_ckks_old.err
HX_ARENA_TRACE=1 timeout 300 python tools/prof_levels.py ckks 64 32 > $out/levels_ckks_new.json 2> $out/levels_ckks_new.err; cat $out/levels_ckks_new.json; grep -c arena $out/levels_ckks_new.err
HX_ARENA_TRACE=1 HX_NORM_OLD=1 timeout 300 python tools/prof_levels.py ckks 64 32 > $out/levels_ckks_old2.json 2> $out/levels_ckks_old2.err; cat $out/levels_ckks_old2.json
timeout 300 python tools/prof_levels.py bgv 128 32 > $out/levels_bgv.json 2> $out/levels_bgv.err; cat $out/levels_bgv.json

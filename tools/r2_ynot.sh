#!/bin/bash
# forward butterfly with Y' = (2x + 4q + 1) + ~X' (HX_YNOT: two v_not_b32 + v_lshl_add_u64 instead of the
# v_sub_co / v_subb_co pair): register-pass microbenchmark, then the default workload on both builds, then the
# transform / mod-switch parity tests on the variant.   gpurun --timeout 185 -- 'bash tools/r2_ynot.sh'
export TMPDIR=/tmp
out=gpurun_out/r2ynot; mkdir -p $out
(./tools/ubench/bfly_new; ./tools/ubench/bfly_ynot; ./tools/ubench/bfly_new; ./tools/ubench/bfly_ynot) > $out/bfly.txt 2>&1; grep -i "forward\|fwd" $out/bfly.txt | cut -c1-150
ROUNDS=1 bash tools/variant_bench.sh default ynot ynot default > /dev/null 2>&1; cp gpurun_out/variants.log $out/variants.log; cat $out/variants.log
HX_LIB=$PWD/helib_amd/lib/variants/libhelib_amd_ynot.so timeout 60 python -m pytest tests -m gpu -q -x -k "ntt or fft or benchmarked_shape or scale_down or bring_to_set or mod_switch" > $out/pytest_ynot.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest_ynot.log

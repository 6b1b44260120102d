#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: the engine and the C++ host built with extra compiler flags into
# helib_amd/lib/variants/NAME/ (libhelib_amd.so + libhelib_amd_host.so side by side, rpath $ORIGIN).  Run a
# benchmark against it with   HX_LIB=$PWD/helib_amd/lib/variants/NAME/libhelib_amd.so
#                             HX_HOST_LIB=$PWD/helib_amd/lib/variants/NAME/libhelib_amd_host.so python bench.py ...
set -e
name=$1; shift
d=helib_amd/lib/variants/$name
mkdir -p $d
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed"
pids=""
for s in ntt_dispatch conv_kernels pfa_kernels rns_mfma_kernels engine; do
  /opt/rocm/bin/hipcc $F "$@" -c helib_amd/csrc/$s.hip -o $d/$s.o & pids="$pids $!"
done
for n in 13 14 15; do   # the row kernels: one translation unit per ring size
  /opt/rocm/bin/hipcc $F "$@" -DHX_NTT_ONLY=$n -c helib_amd/csrc/ntt_kernels.hip -o $d/ntt_kernels_$n.o & pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libhelib_amd.so $d/ntt_kernels_13.o $d/ntt_kernels_14.o $d/ntt_kernels_15.o \
  $d/ntt_dispatch.o $d/conv_kernels.o $d/pfa_kernels.o $d/rns_mfma_kernels.o $d/engine.o
g++ -std=c++17 -O2 -fPIC -shared -Iinclude helib_amd/csrc/host_session.cpp -L$d -lhelib_amd -Wl,-rpath,'$ORIGIN' -o $d/libhelib_amd_host.so
rm -f $d/*.o
ls -la $d

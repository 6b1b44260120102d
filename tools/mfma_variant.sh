#!/bin/bash
# tools/mfma_variant.sh NAME [-DFLAG ...]: only rns_mfma_kernels.hip rebuilt with extra flags, linked with the objects
# of the last full build (helib_amd/lib/*.o) into helib_amd/lib/variants/NAME/ -- seconds instead of minutes.
# Run with HX_LIB / HX_HOST_LIB as tools/build_variant.sh says.
set -e
name=$1; shift
d=helib_amd/lib/variants/$name
L=helib_amd/lib
mkdir -p $d
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed "$@" -c helib_amd/csrc/rns_mfma_kernels.hip -o $d/rns_mfma_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libhelib_amd.so $L/ntt_kernels_13.o $L/ntt_kernels_14.o $L/ntt_kernels_15.o \
  $L/ntt_dispatch.o $L/conv_kernels.o $L/pfa_kernels.o $d/rns_mfma_kernels.o $L/engine.o
g++ -std=c++17 -O2 -fPIC -shared -Iinclude helib_amd/csrc/host_session.cpp -L$d -lhelib_amd -Wl,-rpath,'$ORIGIN' -o $d/libhelib_amd_host.so
rm -f $d/*.o

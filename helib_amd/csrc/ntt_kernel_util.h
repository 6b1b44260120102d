// ntt_kernel_util.h -- launch-shape helpers shared by the row-kernel translation units
// (ntt_kernels.hip, conv_kernels.hip).
#pragma once
#include "dev_common.h"
#include "work_map.h"

// Minimum waves per SIMD requested from the register allocator.  T = N/32
// threads: N=2^13 -> 4 waves/WG, 2^14 -> 8, 2^15 -> 16 (=4 per SIMD already).
// 4 => at most 128 VGPRs, i.e. two 512-thread workgroups per CU for N = 2^14, so one
// row's load / LDS / store phases overlap the other row's butterflies (measured +17 %
// on the forward transform over the 256-VGPR one-workgroup-per-CU build).
#ifndef HX_NTT_MINWAVES
#define HX_NTT_MINWAVES(LOGN) 4
#endif

namespace hx {

// XCD-aware work mapping (work_map.h: a pure function, checked for bijectivity on the host by tests/cpp/work_map_test.cpp)
__device__ __forceinline__ unsigned xcd_remap(unsigned id, unsigned nwg)
{
#ifdef HX_NO_XCD_REMAP
  return id;
#else
  return xcd_remap_id(id, nwg);
#endif
}

// The work-item id, recomputed from the lane counter and the (scalar) wave index each time it is
// asked for.  Every phase gets its own copy: the address arithmetic derived from it then lives
// only inside that phase instead of being kept (and spilled to scratch) across the whole kernel.
// Scratch is what must not happen here: on part of the MI355X pool private-memory accesses are
// slow enough that a dozen spill/reload pairs per thread cost 1.5-3x on these kernels.
__device__ __forceinline__ unsigned fresh_tid(unsigned wave)
{
#ifdef HX_NO_FRESH_TID
  (void)wave;
  return threadIdx.x;
#else
  unsigned lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  return (wave << 6) | lane;
#endif
}
__device__ __forceinline__ unsigned wave_index()
{
  return (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}
// uniform 16-bit table entry as a scalar
__device__ __forceinline__ unsigned uniform_u16(const uint16_t* tab, unsigned i)
{
  return (unsigned)__builtin_amdgcn_readfirstlane((int)tab[i]);
}

}  // namespace hx

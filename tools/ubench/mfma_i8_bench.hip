// mfma_i8_bench.hip -- what one tile of the matrix-core basis extension (helib_amd/csrc/rns_mfma_kernels.hip) costs on
// gfx950, piece by piece: the 2 x steps V_MFMA_I32_32X32X32_I8 of a tile from registers, the four (coefficient, target)
// reductions behind them (recombine eight 24-bit limb sums to 80 bits, one 32-bit quotient estimate, two conditional
// subtractions), and both in one wavefront as the kernel issues them -- at 1, 2 and 3 wavefronts per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I helib_amd/csrc -o mfma_i8_bench tools/ubench/mfma_i8_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#include "mfma_ext.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int STEPS = 10;     // n = 36 sources + the cnt slot
constexpr int TILES = 27;     // 107 targets
constexpr int REPS = 8;       // tile loops per launch

__device__ __forceinline__ uint64_t csub64(uint64_t x, uint64_t m) { return x >= m ? x - m : x; }

// MODE 1: MFMAs only; 2: reductions only; 3: both (the kernel's tile loop without its memory traffic);
// 4: as 3 with the A operand of every step read from the LDS (ds_read_b128) as the kernel does; 5: as 4 with a
// workgroup barrier per tile; 6: MFMAs only, A from the LDS
template <int MODE>
__global__ void __launch_bounds__(256, 2) tile_loop(uint64_t* out, uint64_t q, uint32_t mu80, int tiles)
{
  const unsigned lane = threadIdx.x & 63u;
  v4i b0[STEPS], b1[STEPS], a[STEPS];
#pragma unroll
  for (int j = 0; j < STEPS; j++) {
    b0[j] = v4i{(int)(lane * 0x01010101u + j), (int)(lane * 7u + j), (int)(lane ^ (j * 0x11111111u)), j};
    b1[j] = v4i{(int)(lane * 0x01030507u + j), (int)(lane * 5u + j), (int)(lane ^ (j * 0x01111111u)), j + 1};
    a[j] = v4i{(int)(lane + 3 * j), (int)(lane * 3u + j), (int)(lane ^ (j * 0x10101010u)), j + 2};
  }
  __shared__ v4i a_lds[STEPS * 64];
  constexpr bool LDSA = MODE >= 4;
  constexpr bool DO_MFMA = MODE != 2, DO_RED = MODE != 1 && MODE != 6;
  if (LDSA) {
    for (int j = (int)threadIdx.x; j < STEPS * 64; j += 256)
      a_lds[j] = a[j / 64];
    __syncthreads();
  }
  uint64_t sink = 0;
  v16i init;
#pragma unroll
  for (int r = 0; r < 16; r++)
    init[r] = 5242880 + r;
  for (int rep = 0; rep < REPS; rep++)
    for (int tau = 0; tau < tiles; tau++) {
      v16i acc0 = init, acc1 = init;
      if constexpr (DO_MFMA) {
#pragma unroll
        for (int j = 0; j < STEPS; j++) {
          const v4i aj = LDSA ? a_lds[j * 64 + ((lane + tau) & 63)] : a[j];
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aj, b0[j], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(aj, b1[j], acc1, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r++) {
          acc0[r] = (acc0[r] + tau + (int)sink) & 0xffffff;
          acc1[r] = (acc1[r] + 3 * tau + (int)sink) & 0xffffff;
        }
      }
      if constexpr (DO_RED) {
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
          for (int cb = 0; cb < 2; cb++) {
            const v16i& acc = cb ? acc1 : acc0;
            const uint32_t S[8] = {(uint32_t)acc[8 * s] & 0xffffffu, (uint32_t)acc[8 * s + 1] & 0xffffffu, (uint32_t)acc[8 * s + 2] & 0xffffffu,
                                   (uint32_t)acc[8 * s + 3] & 0xffffffu, (uint32_t)acc[8 * s + 4] & 0xffffffu, (uint32_t)acc[8 * s + 5] & 0xffffffu,
                                   (uint32_t)acc[8 * s + 6] & 0xffffffu, (uint32_t)acc[8 * s + 7] & 0xffffffu};
            const hx::mfx::V80 v = hx::mfx::recombine(S);
            uint64_t r = hx::mfx::red80_lazy(v.lo, v.hi, q, mu80);
            r = csub64(r, q + q);
            r = csub64(r, q);
            sink ^= r;
          }
      } else {
        sink ^= (uint64_t)(uint32_t)acc0[lane & 15] ^ (uint32_t)acc1[(lane + 1) & 15];
      }
      init[0] = (int)(5242880u + (uint32_t)(sink & 0xff));
      if constexpr (MODE == 5)
        __syncthreads();
    }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

template <int MODE>
static int run(const char* name, uint64_t* d, int waves_per_simd)
{
  const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = one per SIMD) x waves_per_simd
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const uint64_t q = (1ull << 59) + 0x12345;
  const uint32_t mu80 = (uint32_t)((((unsigned __int128)1) << 80) / q);
  hipLaunchKernelGGL(tile_loop<MODE>, dim3(blocks), dim3(256), 0, 0, d, q, mu80, TILES);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int it = 0; it < 5; it++) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(tile_loop<MODE>, dim3(blocks), dim3(256), 0, 0, d, q, mu80, TILES);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  const double tiles = (double)REPS * TILES;                       // per wavefront
  const double us_per_tile_per_simd = best * 1e3 / tiles / waves_per_simd;   // SIMD time one wavefront's tile takes
  const double macs = tiles * 2 * STEPS * 32.0 * 32 * 32 * blocks * 4;
  printf("%-28s waves/SIMD %d  %8.1f us   %.3f us per tile and SIMD", name, waves_per_simd, best * 1e3, us_per_tile_per_simd);
  if (MODE != 2)
    printf("   %.0f TOPS", 2 * macs / (best * 1e-3) * 1e-12);
  // the extension of the bits = 6400 digit: 262144 coefficients = 4096 wavefronts of 27 tiles on 1024 SIMDs
  printf("   -> %.1f us per launch of 2^18 coefficients x 107 targets\n", us_per_tile_per_simd * 27 * 4096 / 1024);
  return 0;
}

int main()
{
  uint64_t* d;
  CHECK(hipMalloc(&d, sizeof(uint64_t) * 256 * 4 * 256 * 2));
  for (int w = 1; w <= 4; w++) {
    if (run<1>("mfma only (20 per tile)", d, w)) return 1;
    if (run<2>("4 reductions only", d, w)) return 1;
    if (run<3>("mfma + 4 reductions", d, w)) return 1;
    if (run<6>("mfma only, A from LDS", d, w)) return 1;
    if (run<4>("mfma (A from LDS) + 4 red.", d, w)) return 1;
    if (run<5>("... + barrier per tile", d, w)) return 1;
  }
  (void)hipFree(d);
  return 0;
}

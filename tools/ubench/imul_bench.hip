// imul_bench.hip -- integer-multiply issue-rate microbenchmark for gfx950:
// how fast can the vector ALU do the 32/64-bit multiplies a 60-bit Shoup butterfly needs?
// Build: hipcc --offload-arch=gfx950 -O3 -o imul_bench imul_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 4096;
constexpr int ILP = 8;

template <int MODE>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint64_t seed, uint64_t q)
{
  uint64_t a[ILP], b = seed | 1;
  for (int i = 0; i < ILP; i++) a[i] = seed * (threadIdx.x + 1 + i * 977) + blockIdx.x;
  uint32_t b32 = (uint32_t)b;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < ILP; i++) {
      if (MODE == 0) {  // v_mul_lo_u32
        a[i] = (uint32_t)a[i] * b32 + 1;
      } else if (MODE == 1) {  // v_mul_hi_u32
        a[i] = __umulhi((uint32_t)a[i], b32) + it;
      } else if (MODE == 2) {  // v_mad_u64_u32
        a[i] = (uint64_t)(uint32_t)a[i] * b32 + a[i];
      } else if (MODE == 3) {  // 64-bit mulhi
        a[i] = __umul64hi(a[i], b) + it;
      } else if (MODE == 4) {  // 64-bit mullo
        a[i] = a[i] * b + it;
      } else if (MODE == 5) {  // Shoup lazy modmul: mulhi64 + 2 mullo64
        uint64_t h = __umul64hi(a[i], b);
        a[i] = a[i] * seed - h * q;
      } else if (MODE == 6) {  // v_mul_u32_u24
        a[i] = __umul24((uint32_t)a[i], b32) + 1;
      } else if (MODE == 7) {  // 64-bit add (baseline full-rate op pair)
        a[i] = a[i] + b;
      } else if (MODE == 8) {  // f64 fma
        double d = __longlong_as_double(a[i]);
        d = __fma_rn(d, 1.0000001, 0.5);
        a[i] = __double_as_longlong(d);
      }
    }
  }
  uint64_t s = 0;
  for (int i = 0; i < ILP; i++) s ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, uint64_t* d)
{
  int blocks = 256 * 8;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0x9E3779B97F4A7C15ull, (1ull << 59) + 12345);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0x9E3779B97F4A7C15ull, (1ull << 59) + 12345);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  double ops = (double)blocks * 256 * ITER * ILP;
  printf("%-28s %8.3f ms  %9.1f Gop/s  (%.2f cycles/wave-op/SIMD @2.4GHz)\n", name, ms, ops / ms / 1e6,
         (256.0 * 4 * 2.4e9) / (ops / 64 / (ms * 1e-3)));
  return 0;
}

int main()
{
  uint64_t* d;
  CHECK(hipMalloc(&d, 256 * 8 * 256 * 8));
  run<7>("add_u64", d);
  run<0>("mul_lo_u32", d);
  run<1>("mul_hi_u32", d);
  run<2>("mad_u64_u32", d);
  run<6>("mul_u32_u24", d);
  run<3>("mulhi_u64", d);
  run<4>("mullo_u64", d);
  run<5>("shoup_lazy (hi64+2*lo64)", d);
  run<8>("fma_f64", d);
  return 0;
}

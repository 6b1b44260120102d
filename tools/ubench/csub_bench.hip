// csub_bench.hip -- conditional subtraction x in [0,2m) -> [0,m), m <= 2^63, three ways; each is
// checked against (x >= m ? x - m : x) on random inputs and timed (wall clock) at 8 waves / SIMD.
//   A  subtract, select on the borrow (ntt_core.h round 1): v_sub_co, v_subb_co, 2 x v_cndmask
//   C  d = x - m (one v_lshl_add_u64 with -m); no borrow <=> d.hi < x.hi: v_cmp, v_min_u32 (hi), v_cndmask (lo)
//   D  d = x - m; borrow <=> d negative as a signed number: d + (m & (d.hi >> 31)), no VCC at all
//   hipcc --offload-arch=gfx950 -O3 -o csub_bench csub_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2048;

template <int MODE>
__device__ __forceinline__ uint64_t csub(uint64_t x, uint64_t m, uint64_t negm)
{
  const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), ml = (uint32_t)m, mh = (uint32_t)(m >> 32);
  if constexpr (MODE == 0) {
    uint32_t dl, dh;
    asm("v_sub_co_u32 %0, vcc, %2, %4\n\ts_nop 1\n\tv_subb_co_u32 %1, vcc, %3, %5, vcc\n\ts_nop 1\n\t"
        "v_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc"
        : "=&v"(dl), "=&v"(dh) : "v"(xl), "v"(xh), "v"(ml), "v"(mh) : "vcc");
    return ((uint64_t)dh << 32) | dl;
  } else if constexpr (MODE == 1) {
    uint64_t d;
    uint32_t rl, rh;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(x), "s"(negm));
    const uint32_t dl = (uint32_t)d, dh = (uint32_t)(d >> 32);
    asm("v_cmp_lt_u32 vcc, %2, %3\n\tv_min_u32 %1, %2, %3\n\ts_nop 0\n\tv_cndmask_b32 %0, %4, %5, vcc"
        : "=&v"(rl), "=&v"(rh) : "v"(dh), "v"(xh), "v"(xl), "v"(dl) : "vcc");
    return ((uint64_t)rh << 32) | rl;
  } else {
    uint64_t d, r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(x), "s"(negm));
    uint32_t msk, tl, th;
    asm("v_ashrrev_i32 %0, 31, %1" : "=v"(msk) : "v"((uint32_t)(d >> 32)));
    asm("v_and_b32 %0, %1, %2" : "=v"(tl) : "s"(ml), "v"(msk));
    asm("v_and_b32 %0, %1, %2" : "=v"(th) : "s"(mh), "v"(msk));
    const uint64_t t = ((uint64_t)th << 32) | tl;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(d), "v"(t));
    return r;
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint64_t seed, uint64_t m, int check)
{
  uint64_t a[8];
  const uint64_t negm = 0 - m;
  for (int i = 0; i < 8; i++) {
    uint64_t z = seed * (blockIdx.x * 256 + threadIdx.x + 1 + i * 977) + 0x9E3779B97F4A7C15ull * i;
    z ^= z >> 29;
    a[i] = z % (2 * m - 1);
  }
  if (check) {
    uint64_t bad = 0;
    for (int i = 0; i < 8; i++) {
      // also the edges
      uint64_t xs[4] = {a[i], m - 1, m, (i & 1) ? 2 * m - 1 : 0};
      for (int j = 0; j < 4; j++) {
        uint64_t x = xs[j], want = x >= m ? x - m : x;
        bad |= (csub<MODE>(x, m, negm) != want);
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = bad;
    return;
  }
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      a[i] = csub<MODE>(a[i] + (a[(i + 1) & 7] & 0xffff), m, negm);  // (+ one add so that values keep moving)
  }
  uint64_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, uint64_t* d, uint64_t m)
{
  int blocks = 256 * 8;
  std::vector<uint64_t> h((size_t)blocks * 256);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0x9E3779B97F4A7C15ull, m, 1);
  CHECK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
  uint64_t bad = 0;
  for (auto v : h) bad |= v;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0x9E3779B97F4A7C15ull, m, 0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 0x9E3779B97F4A7C15ull, m, 0);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-60s m=%016llx  %s  %.3f ms  %.2f ns per csub(+add) per SIMD\n", name, (unsigned long long)m,
         bad ? "WRONG" : "exact", ms, ms * 1e6 / ((double)ITER * 8 * 8));
  return 0;
}

int main()
{
  uint64_t* d;
  CHECK(hipMalloc(&d, (size_t)256 * 8 * 256 * 8));
  for (uint64_t m : std::vector<uint64_t>{8 * 1152921504606584833ull, 1152921504606584833ull, 4 * ((1ull << 40) - 87),
                     1ull << 63}) {
    run<0>("A sub_co/subb_co + 2 cndmask (borrow select)", d, m);
    run<1>("C lshl_add(-m) + cmp hi + min hi + cndmask lo", d, m);
    run<2>("D lshl_add(-m) + ashr + 2 and + lshl_add (no VCC)", d, m);
  }
  return 0;
}

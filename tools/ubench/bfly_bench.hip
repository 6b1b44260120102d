// bfly_bench.hip -- the register passes of the row NTT alone (no LDS, no global traffic): cycles
// per butterfly of run_pass<5> forward / inverse with wave-uniform twiddles, at 1..4 waves per SIMD
// of 512-thread workgroups (the N = 2^14 kernel's shape).  Build twice to compare butterflies:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bfly_new bfly_bench.hip
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHX_SHOUP4_OLD -o bfly_old bfly_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../helib_amd/csrc/ntt_core.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using namespace hx;
constexpr int ITER = 64;

template <bool INV, class AR>
__global__ void __launch_bounds__(512, 4) k(uint64_t* out, uint64_t* cyc, const typename AR::Tw* tw, uint64_t q)
{
  uint64_t v[32];
  const QC c = make_qc(q);
  for (int e = 0; e < 32; e++)
    v[e] = (q >> 1) + (uint64_t)threadIdx.x * 977u + (uint64_t)e * 131071u + blockIdx.x;
  const uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < ITER; it++) {
    if constexpr (!INV)
      run_pass<AR, 5, false, 1, 31, AR::U>(v, c, [&](int, int sp, int kk, uint32_t) { return tw[(1 << sp) - 1 + kk]; });
    else
      run_pass<AR, 5, true, 1, 31, 1>(v, c, [&](int, int sp, int kk, uint32_t) { return tw[(1 << sp) - 1 + kk]; });
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint64_t s = 0;
  for (int e = 0; e < 32; e++) s ^= v[e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <bool INV, class AR>
int run(const char* name, uint64_t* d, uint64_t* dc, const typename AR::Tw* tw, uint64_t q)
{
  printf("%-28s", name);
  for (int wgs : {1, 2}) {  // workgroups per CU: 512 threads = 2 waves per SIMD each
    int blocks = 256 * wgs;
    hipLaunchKernelGGL((k<INV, AR>), dim3(blocks), dim3(512), 0, 0, d, dc, tw, q);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<INV, AR>), dim3(blocks), dim3(512), 0, 0, d, dc, tw, q);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h((size_t)blocks * 8);
    CHECK(hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= (double)h.size();
    const double bf = (double)ITER * 80;  // butterflies per thread: 5 stages x 16
    const int wps = 2 * wgs;
    // ns per wave-butterfly per SIMD: blocks x 8 waves over 1024 SIMDs
    const double ns = ms * 1e6 / (bf * (double)blocks * 8.0 / 1024.0);
    printf("  %d waves/SIMD: %6.1f cycles/butterfly/SIMD (wave sees %6.1f), %.3f ms = %5.2f ns/butterfly/SIMD, clock %.2f GHz", wps,
           avg / bf / wps, avg / bf, ms, ns, avg / (ms * 1e6));
  }
  printf("\n");
  return 0;
}

int main()
{
  const uint64_t q = 1152921504606584833ull - 0;  // placeholder, replaced below
  (void)q;
  // a 60-bit chain prime of m = 32768 (PrimeGenerator(60, 32768)): q = 0xed0000000000001 = 237 * 2^52 + 1
  const uint64_t qp = 0xed0000000000001ull;
  std::vector<TW> t(32);
  std::vector<TWM> tm(32);
  for (int i = 0; i < 32; i++) {
    t[i].w = (qp / 3) + (uint64_t)i * 0x9E3779B97F4A7ull % (qp / 2);
    t[i].wp = (uint64_t)((((unsigned __int128)t[i].w) << 64) / qp);
    tm[i] = tw_mont_form(t[i].w, qp);
  }
  uint64_t *d, *dc;
  TW* dt;
  TWM* dtm;
  CHECK(hipMalloc(&d, (size_t)512 * 512 * 8));
  CHECK(hipMalloc(&dc, (size_t)512 * 8 * 8));
  CHECK(hipMalloc(&dt, 32 * sizeof(TW)));
  CHECK(hipMalloc(&dtm, 32 * sizeof(TWM)));
  CHECK(hipMemcpy(dt, t.data(), 32 * sizeof(TW), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dtm, tm.data(), 32 * sizeof(TWM), hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; rep++) {
    run<false, ArShoup>("forward Shoup (shoup4_acc)", d, dc, dt, qp);
    run<false, ArProth>("forward Proth (mont_acc)", d, dc, dtm, qp);
    run<true, ArShoup>("inverse Shoup", d, dc, dt, qp);
    run<true, ArProth>("inverse Proth", d, dc, dtm, qp);
  }
  return 0;
}

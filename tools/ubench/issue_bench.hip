// issue_bench.hip -- what do the instructions of the 60-bit butterfly cost on gfx950?
// Every loop body is inline asm (nothing for the compiler to fold), timed with the shader clock
// (s_memtime) inside the kernel, so the result is cycles per wave-instruction per SIMD whatever the
// clock is; the wall time gives the clock itself.  W waves per SIMD (1, 2, 4, 8) x ILP registers.
//   hipcc --offload-arch=gfx950 -O3 -o issue_bench issue_bench.hip && ./issue_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITER = 8192;   // long enough that the launch overhead inside the event pair is < 2 % at one wave per SIMD

// 8 independent instructions per loop iteration, or a dependent chain (DEP)
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int MODE>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint64_t* cyc, uint64_t seed)
{
  uint64_t a[8];
  uint32_t x[8], y[8];
  for (int i = 0; i < 8; i++) {
    a[i] = seed * (threadIdx.x + 1 + i * 977) + blockIdx.x;
    x[i] = (uint32_t)(a[i] >> 7) | 1u;
    y[i] = (uint32_t)(a[i] >> 23) | 3u;
  }
  uint64_t sg = seed | 5;  // uniform
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; it++) {
    if constexpr (MODE == 0) {  // v_mad_u64_u32, independent accumulators
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x[i]), "v"(y[i]) : "vcc");
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 1) {  // v_mad_u64_u32, one dependent chain
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[0]) : "v"(x[i]), "v"(y[i]) : "vcc");
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 2) {  // v_mul_lo_u32
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 3) {  // v_mul_hi_u32
#define OP(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 4) {  // v_lshl_add_u64
#define OP(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a[i]) : "s"(sg));
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 5) {  // v_add_u32
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 6) {  // 64-bit subtract with borrow through VCC (two instructions + wait state)
#define OP(i)                                                                  \
  {                                                                            \
    uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32);                 \
    asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n\ts_nop 1\n\tv_subb_co_u32 %1, vcc, %1, %3, vcc" \
                 : "+v"(lo), "+v"(hi) : "v"(x[i]), "v"(y[i]) : "vcc");         \
    a[i] = ((uint64_t)hi << 32) | lo;                                          \
  }
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 7) {  // v_cndmask_b32 (mask in vcc)
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(y[i]) : "vcc");
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 8) {  // v_fma_f64
      double d[8];
#define OP(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(sg));
      REP8(OP)
#undef OP
      (void)d;
    } else if constexpr (MODE == 9) {  // v_mad_u64_u32 with an SGPR multiplier
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x[i]), "s"((uint32_t)sg) : "vcc");
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 10) {  // v_mad_u64_u32 x 1 (the "add a 32-bit value into 64 bits" form)
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(a[i]) : "v"(x[i]) : "vcc");
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 11) {  // v_mov_b32
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(y[i]));
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 12) {  // v_add3_u32
#define OP(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y[i]), "v"(y[(i + 1) & 7]));
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 14) {  // 64-bit add with carry through VCC (v_add_co + v_addc_co, per pair)
#define OP(i)                                                                  \
  {                                                                            \
    uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32);                 \
    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\ts_nop 1\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" \
                 : "+v"(lo), "+v"(hi) : "v"(x[i]), "v"(y[i]) : "vcc");         \
    a[i] = ((uint64_t)hi << 32) | lo;                                          \
  }
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 15) {  // the conditional subtraction as the kernels write it: sub, subb, two selects
#define OP(i)                                                                  \
  {                                                                            \
    uint32_t lo = (uint32_t)a[i], hi = (uint32_t)(a[i] >> 32), tl, th;         \
    asm volatile("v_sub_co_u32 %2, vcc, %0, %4\n\ts_nop 1\n\tv_subb_co_u32 %3, vcc, %1, %5, vcc\n\ts_nop 1\n\t" \
                 "v_cndmask_b32 %0, %2, %0, vcc\n\tv_cndmask_b32 %1, %3, %1, vcc"            \
                 : "+v"(lo), "+v"(hi), "=&v"(tl), "=&v"(th) : "v"(x[i]), "v"(y[i]) : "vcc");   \
    a[i] = ((uint64_t)hi << 32) | lo;                                          \
  }
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 16) {  // v_and_b32 (any simple 32-bit op)
#define OP(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y[i]));
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 17) {  // v_alignbit_b32
#define OP(i) asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(x[i]) : "v"(y[i]));
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 18) {  // v_lshlrev_b64
#define OP(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[i]));
      REP8(OP)
#undef OP
    } else if constexpr (MODE == 19) {  // ds_write_b32 + ds_read_b32 (per pair, private slot, waits at the end of the group)
      __shared__ uint32_t sh[256 * 9];
#define OP(i) asm volatile("ds_write_b32 %1, %0\n\tds_read_b32 %0, %1" : "+v"(x[i]) : "v"((threadIdx.x * 9 + i) * 4) : "memory");
      REP8(OP)
#undef OP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      (void)sh;
    } else if constexpr (MODE == 20) {  // v_cvt_f64_u32
      double dd;
#define OP(i) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(x[i]));
      REP8(OP)
#undef OP
      (void)dd;
    } else if constexpr (MODE == 13) {  // mad_u64_u32 alternating with add_u32 (does a cheap op hide in the multiplier's shadow?)
#define OP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_add_u32 %1, %1, %2" : "+v"(a[i]), "+v"(x[i]) : "v"(y[i]) : "vcc");
      REP8(OP)
#undef OP
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint64_t s = 0;
  for (int i = 0; i < 8; i++) s ^= a[i] ^ x[i] ^ y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
int run(const char* name, int per_instr, uint64_t* d, uint64_t* dc)
{
  printf("%-44s", name);
  for (int wps : {1, 2, 4, 8}) {          // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
    int blocks = 256 * wps;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, 0x9E3779B97F4A7C15ull);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, dc, 0x9E3779B97F4A7C15ull);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<uint64_t> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= (double)h.size();
    // cycles (shader clock) one wave spends per instruction; with wps waves sharing a SIMD the SIMD
    // issues one such instruction every avg / (ITER*8*per_instr) / wps cycles
    double per_wave = avg / ((double)ITER * 8 * per_instr);
    // wall-clock view, independent of what the in-kernel counter counts: nanoseconds a SIMD needs per wave-instruction
    // (two launches are timed: the second one's event pair; a launch overhead of ~5 us is inside ms)
    const double ns = (double)ms * 1e6 / ((double)ITER * 8 * per_instr * wps);
    printf("  w%d: %6.2f cyc/instr/SIMD (wave sees %6.2f; %.3f ms = %5.2f ns/instr/SIMD)", wps, per_wave / wps, per_wave, ms, ns);
  }
  printf("\n");
  return 0;
}

int main()
{
  uint64_t *d, *dc;
  CHECK(hipMalloc(&d, (size_t)256 * 8 * 256 * 8));
  CHECK(hipMalloc(&dc, (size_t)256 * 8 * 4 * 8));
  run<0>("v_mad_u64_u32 (8 independent)", 1, d, dc);
  run<1>("v_mad_u64_u32 (dependent chain)", 1, d, dc);
  run<9>("v_mad_u64_u32 (SGPR multiplier)", 1, d, dc);
  run<10>("v_mad_u64_u32 x 1", 1, d, dc);
  run<2>("v_mul_lo_u32", 1, d, dc);
  run<3>("v_mul_hi_u32", 1, d, dc);
  run<4>("v_lshl_add_u64", 1, d, dc);
  run<5>("v_add_u32", 1, d, dc);
  run<12>("v_add3_u32", 1, d, dc);
  run<11>("v_mov_b32", 1, d, dc);
  run<6>("v_sub_co + s_nop 1 + v_subb_co (per pair)", 1, d, dc);
  run<7>("v_cndmask_b32", 1, d, dc);
  run<8>("v_fma_f64", 1, d, dc);
  run<13>("v_mad_u64_u32 + v_add_u32 (per pair)", 1, d, dc);
  run<14>("v_add_co + s_nop 1 + v_addc_co (per pair)", 1, d, dc);
  run<15>("csub: sub, subb, 2 x cndmask (per group)", 1, d, dc);
  run<16>("v_and_b32", 1, d, dc);
  run<17>("v_alignbit_b32", 1, d, dc);
  run<18>("v_lshlrev_b64", 1, d, dc);
  run<19>("ds_write_b32 + ds_read_b32 (per pair)", 1, d, dc);
  run<20>("v_cvt_f64_u32", 1, d, dc);
  return 0;
}

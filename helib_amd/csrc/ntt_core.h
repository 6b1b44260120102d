// ntt_core.h -- negacyclic NTT of one DoubleCRT row (m = 2^k, N = m/2), written
// for gfx950: one workgroup owns one row, 32 coefficients per thread live in
// VGPRs, three register passes (radix-32, radix-32, radix-2^LC) with two LDS
// transposes between them.  Replaces Cmodulus::FFT_aux / iFFT power-of-two
// branches (HElib src/CModulus.cpp:362-429, :493-553) including NTL's
// FFTFwd/FFTRev1, the powers[] pre-twist and the BitReverseCopy passes.
//
// The phase functions are HOST+DEVICE so that tests/ can replay the exact
// index arithmetic of the kernel thread-by-thread on the CPU (no GPU in the
// build container).  The product path only ever runs them inside the HIP
// kernels of ntt_kernels.hip.
//
// Math: y[j] = sum_i x_i * psi^(i*(2j+1)), psi = w0 a
// primitive 2N-th root of unity, j natural order.  Forward = Cooley-Tukey
// butterflies with psi powers merged into the twiddles (bit-reversed table),
// natural in -> bit-reversed positions, un-reversed for free by the store
// addressing of the last pass.  Inverse = the mirrored Gentleman-Sande network
// with N^-1 folded into the last stage.
//
// Lazy arithmetic of the row kernels (q < 2^60, i.e. 16q <= 2^64, is REQUIRED; HElib never
// makes larger primes, src/macro.h:21): the twiddle product is Shoup's with an APPROXIMATE
// high product (shoup4: the quotient estimate may be up to 2 short, result in [0,4q) for any
// 64-bit input; 3 instead of 4 partial products and no carry chain), and the conditional
// subtraction of the untouched butterfly arm is done only where a compile-time bound
// (in units of q, tracked per stage / per element below) would pass 16q.  The small-ring
// kernel keeps the classical Harvey butterflies ([0,4q) forward / [0,2q) inverse).
#pragma once
#include <stdint.h>
#include <utility>
#if defined(HX_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#endif

#if defined(__HIPCC__)
#define HXD __host__ __device__ __forceinline__
#else
#define HXD inline
#endif

namespace hx {

struct TW {
  uint64_t w;   // twiddle in [0,q)
  uint64_t wp;  // floor(w * 2^64 / q)   (NTL::PrepMulModPrecon analogue)
};

HXD uint64_t mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

// x * w mod q, result in [0, 2q) for ANY 64-bit x (Shoup / Harvey).
HXD uint64_t shoup_lazy(uint64_t x, TW t, uint64_t q)
{
  uint64_t h = mulhi64(x, t.wp);
  return x * t.w - h * q;
}

HXD uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// Optional run-time check of the compile-time bounds (CPU replay only).
#if defined(HX_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#define HX_BOUND(x, B, q)                                                              \
  do {                                                                                 \
    if ((unsigned __int128)(x) >= (unsigned __int128)(B) * (q)) {                      \
      std::fprintf(stderr, "bound violated at %s:%d (B=%d)\n", __FILE__, __LINE__, (int)(B)); \
      std::abort();                                                                    \
    }                                                                                  \
  } while (0)
#else
#define HX_BOUND(x, B, q) ((void)0)
#endif

// per-row constants of the lazy scheme
struct QC {
  uint64_t q;
  uint64_t nq;  // 2^64 - q
  uint64_t q4, q8;
  uint32_t mu32;  // floor(2^64 / q) when that fits 32 bits (q > 2^32), else 0
  // Proth-form rows only (q = qh 2^32 + 1, see ArProth below; dead code elsewhere)
  uint32_t qh;    // q >> 32
  uint64_t c1;    // (1 + qh)(2^32 + 1): the two carry constants of the word-wise reduction
  uint64_t q2;    // 2q
};
// mu64 = floor(2^64 / q) (PrimeDev::mu64 on the device)
HXD QC make_qc(uint64_t q, uint64_t mu64)
{
  QC c;
  c.q = q;
  c.nq = 0 - q;
  c.q4 = q << 2;
  c.q8 = q << 3;
  c.mu32 = (mu64 >> 32) ? 0u : (uint32_t)mu64;
  c.qh = (uint32_t)(q >> 32);
  c.c1 = ((uint64_t)(c.qh + 1u) << 32) + (c.qh + 1u);
  c.q2 = q << 1;
  return c;
}
// q = t 2^s + 1 with s >= 32 (every prime src/PrimeGenerator.h:66-118 makes for the rings and sizes of
// the benchmark chains: cand = ((t*m) << k) + 1): the rows the Proth-form butterflies below serve
HXD bool is_proth32(uint64_t q) { return (uint32_t)q == 1u && (q >> 32) != 0 && (q >> 60) == 0; }
// floor(2^64/q) = floor((2^64-1)/q) for odd q > 1
HXD QC make_qc(uint64_t q) { return make_qc(q, ~(uint64_t)0 / q); }

// y * w mod q as a value in [0, 4q), for ANY 64-bit y.  With wp = floor(w 2^64 / q) and
// h = floor(y wp / 2^64), Shoup's r = y w - h q lies in [0, 2q).  Here h is replaced by
//   h' = yh*ph + hi32(yh*pl) + hi32(yl*ph)      (the yl*pl term and the carries of the two
// middle products' low halves are dropped), h - 2 <= h' <= h, so r' = r + (h-h') q < 4q; all of
// it modulo 2^64, which is exact because r' < 2^62.  -h' q is accumulated as + h' * (2^64 - q)
// so that the low 64 bits form one multiply-add chain.
//
// Device form (shoup4_acc): x + y*w - h'q for a 64-bit addend x, everything as v_mad_u64_u32
// chains.  On gfx950 a v_mad_u64_u32 issues in 5.05 cycles per wave64 against 9.12 for
// v_mul_hi_u32 and 5.33 + an add for v_mul_lo_u32 (profiles/r01_imul_issue_rates.txt), and its
// 64-bit addend is free: the two high halves come from full products, the four cross products
// of the low 64 bits are one accumulation chain whose low word is all that is used, and the
// butterfly's x + T rides on the chain's first addend.  The empty asm statements keep the
// compiler from narrowing the chains back into v_mul_hi / v_mul_lo + add.  Same h', same value.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HX_SHOUP4_OLD)
#define HX_KEEP64(x) asm("" : "+v"(x))
#else
#define HX_KEEP64(x) ((void)0)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint64_t mad_x1(uint32_t a, uint64_t c)  // c + a
{
  uint64_t d, cy;
  asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(cy) : "v"(a), "v"(c));
  return d;
}
#endif
HXD uint64_t shoup4_acc(uint64_t y, TW t, uint64_t nq, uint64_t x)
{
  const uint32_t yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
  const uint32_t pl = (uint32_t)t.wp, ph = (uint32_t)(t.wp >> 32);
  const uint32_t wl = (uint32_t)t.w, wh = (uint32_t)(t.w >> 32);
  const uint32_t nl = (uint32_t)nq, nh = (uint32_t)(nq >> 32);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HX_SHOUP4_OLD)
#ifndef HX_SHOUP4_MADX1
  // the two high halves by v_mul_hi_u32 and summed by the compiler: measured 1-2.5 % faster on
  // the row kernels than the all-multiply-add form below (a v_mad_u64_u32 issues in 2.4 ns per
  // wave64 per SIMD, a v_mul_hi_u32 in 1.9 ns: tools/ubench/issue_bench.hip, profiles/r02_issue_rates.txt)
  const uint32_t a1 = mulhi32(yh, pl), b1 = mulhi32(yl, ph);
  const uint64_t h = (uint64_t)yh * ph + a1 + b1;
#else
  uint64_t A = (uint64_t)yh * pl, B = (uint64_t)yl * ph;
  HX_KEEP64(A);
  HX_KEEP64(B);
  // + hi32(A) + hi32(B) as multiply-adds by 1: a 32-bit value enters a 64-bit sum without a
  // (value, 0) register pair having to be built for it (two v_mov each otherwise)
  uint64_t h = (uint64_t)yh * ph;
  h = mad_x1((uint32_t)(A >> 32), h);
  h = mad_x1((uint32_t)(B >> 32), h);
#endif
  const uint32_t hl = (uint32_t)h, hh = (uint32_t)(h >> 32);
  uint64_t u = (uint64_t)yl * wh;
  u += (uint64_t)yh * wl;
  u += (uint64_t)hl * nh;
  u += (uint64_t)hh * nl;
  HX_KEEP64(u);
  uint64_t acc = (uint64_t)yl * wl + x;
  HX_KEEP64(acc);  // (x stays the first multiply-add's addend: re-associated, it came back as a 64-bit add of its own)
  acc += (uint64_t)hl * nl;
  const uint32_t al = (uint32_t)acc, ah = (uint32_t)(acc >> 32);
  uint32_t rh;
  asm("v_add_u32 %0, %1, %2" : "=v"(rh) : "v"(ah), "v"((uint32_t)u));
  return ((uint64_t)rh << 32) | al;
#else
  const uint32_t a1 = mulhi32(yh, pl), b1 = mulhi32(yl, ph);
  const uint64_t h = (uint64_t)yh * ph + a1 + b1;
  const uint32_t hl = (uint32_t)h, hh = (uint32_t)(h >> 32);
  const uint32_t t0 = yl * wh + yh * wl + hl * nh + hh * nl;
  uint64_t acc = (uint64_t)yl * wl + x;
  acc += (uint64_t)hl * nl;
  const uint32_t al = (uint32_t)acc, ah = (uint32_t)(acc >> 32);
  uint32_t rh;
#if defined(__HIP_DEVICE_COMPILE__)
  // kept opaque: otherwise the high-word sum is re-associated into a 64-bit shift+add pair
  asm("v_add_u32 %0, %1, %2" : "=v"(rh) : "v"(ah), "v"(t0));
#else
  rh = ah + t0;
#endif
  return ((uint64_t)rh << 32) | al;
#endif
}
HXD uint64_t shoup4(uint64_t y, TW t, uint64_t nq) { return shoup4_acc(y, t, nq, 0); }

// ---- Proth-form Montgomery product (round 5) -----------------------------------------------------
// Every prime of a HElib chain is q = t 2^s + 1 (src/PrimeGenerator.h:66-118), and for the rings and
// prime sizes of the benchmark chains s >= 32: the low word of q is 1.  Then q^-1 = 1 (mod 2^32) and a
// word-wise Montgomery reduction needs NO multiplication for its quotient digit: with W = w 2^64 mod q
// and T = y W, taking n = ~lo32(T) (so that n + 1 = -T mod 2^32, and the low word of T + (n+1) q is
// 2^32 exactly: the carry out of it is the CONSTANT 1, no data-dependent borrow)
//      (T + (n+1) q) / 2^32  =  (T >> 32) + 1 + (n+1) qh,        qh = q >> 32,
// twice.  Six 32x32 multiply-adds in all (four for T, one n*qh per word) against the seven + two
// v_mul_hi_u32 of shoup4_acc, no quotient estimate, and an 8-byte table entry instead of 16.
//   a = yl wl                                   n0 = ~lo32(a)
//   G = yl wh + n0 qh + hi32(a) + yh wl + c1    n1 = ~lo32(G)       c1 = (1+qh)(2^32+1): both carry constants
//   D = yh wh + n1 qh + hi32(G) + x
// D = x + R with R = (y W + M q) / 2^64, M = (n0+1) + (n1+1) 2^32 <= 2^64 + 2^32:  R = y w (mod q) and
//   0 < R < q (1 + y W / (q 2^64) + 2^-32) <= q (1 + By/16 + 2^-32)   for y < By q, q < 2^60, W < q.
// No 64-bit sum may wrap: G < 2^64 needs yh wl < 2^64 - 3 2^60 - 2^34, i.e. y < 13 2^60 - 2^34 -- the
// multiplied operand must stay below 12.9375 q (207/16; bounds of the Proth butterflies are tracked in
// sixteenths of q), and D needs x + R < 2^64.  The two 32-bit high words enter their 64-bit sums as
// multiply-adds by 1 (mad_x1): no (value, 0) register pair to build.  Exact for every y below the limit,
// checked against 128-bit arithmetic in tests/test_host_logic.py and replayed in tests/cpp/ntt_replay.cpp.
typedef uint64_t TWM;  // W = w 2^64 mod q
HXD uint64_t mont_acc(uint64_t y, TWM W, const QC& c, uint64_t x)
{
  const uint32_t yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
  const uint32_t wl = (uint32_t)W, wh = (uint32_t)(W >> 32);
  uint64_t a = (uint64_t)yl * wl;
  HX_KEEP64(a);
  const uint32_t n0 = ~(uint32_t)a;
  uint64_t G = (uint64_t)yl * wh + c.c1;
  G += (uint64_t)n0 * c.qh;
  HX_KEEP64(G);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HX_MONT_PLAINADD)
  G = mad_x1((uint32_t)(a >> 32), G);
#else
  G += (uint32_t)(a >> 32);
#endif
  G += (uint64_t)yh * wl;
  HX_KEEP64(G);
  const uint32_t n1 = ~(uint32_t)G;
  uint64_t D = (uint64_t)yh * wh + x;
  HX_KEEP64(D);
  D += (uint64_t)n1 * c.qh;
  HX_KEEP64(D);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HX_MONT_PLAINADD)
  D = mad_x1((uint32_t)(G >> 32), D);
#else
  D += (uint32_t)(G >> 32);
#endif
  return D;
}
HXD uint64_t mont_mul(uint64_t y, TWM W, const QC& c) { return mont_acc(y, W, c, 0); }
// The reduction alone, for a 128-bit value T = lo + hi 2^64 that is already there (a product of two data words, or a
// sum of two): (T + M q) / 2^64 = T 2^-64 (mod q), in (0, T / 2^64 + q (1 + 2^-32)); needs hi < 2^62.  Two
// multiply-adds against the seven multiplications and up to three conditional subtractions of a Barrett reduction
// of the same value; the stray 2^-64 is folded into the constant the value is multiplied by next.
HXD uint64_t mont_redc128(uint64_t lo, uint64_t hi, const QC& c)
{
  const uint32_t n0 = ~(uint32_t)lo;
  uint64_t G = (uint64_t)n0 * c.qh + c.c1;
  HX_KEEP64(G);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HX_MONT_PLAINADD)
  G = mad_x1((uint32_t)(lo >> 32), G);
#else
  G += (uint32_t)(lo >> 32);
#endif
  const uint32_t n1 = ~(uint32_t)G;
  uint64_t D = (uint64_t)n1 * c.qh + hi;
  HX_KEEP64(D);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HX_MONT_PLAINADD)
  D = mad_x1((uint32_t)(G >> 32), D);
#else
  D += (uint32_t)(G >> 32);
#endif
  return D;
}
// x in [0, 2m) -> [0, m)
HXD uint64_t csub(uint64_t x, uint64_t m)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HX_CSUB_PLAIN)
  // subtract, then select on the borrow: 4 instructions (the compiler's own form compares first: 5).
  // gfx950 needs two wait states between a VALU write of VCC and a VALU read of it as carry-in or
  // select mask; the compiler's hazard recogniser does not look inside inline asm, hence the s_nops.
  const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), ml = (uint32_t)m, mh = (uint32_t)(m >> 32);
  uint32_t dl, dh;
  asm("v_sub_co_u32 %0, vcc, %2, %4\n\t"
      "s_nop 1\n\t"
      "v_subb_co_u32 %1, vcc, %3, %5, vcc\n\t"
      "s_nop 1\n\t"
      "v_cndmask_b32 %0, %0, %2, vcc\n\t"
      "v_cndmask_b32 %1, %1, %3, vcc"
      : "=&v"(dl), "=&v"(dh)
      : "v"(xl), "v"(xh), "v"(ml), "v"(mh)
      : "vcc");
  return ((uint64_t)dh << 32) | dl;
#else
  return x >= m ? x - m : x;
#endif
}

HXD constexpr unsigned brev_bits(unsigned x, int bits)
{
  unsigned r = 0;
  for (int i = 0; i < bits; i++)
    r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}
HXD unsigned brev5(unsigned x)
{
  return ((x & 1u) << 4) | ((x & 2u) << 2) | (x & 4u) | ((x & 8u) >> 2) | ((x & 16u) >> 4);
}
HXD unsigned brev10(unsigned x) { return (brev5(x & 31u) << 5) | brev5(x >> 5); }

// Geometry for N = 2^LOGN, LOGN in {13,14,15}.
template <int LOGN>
struct Geo {
  static constexpr int LC = LOGN - 10;       // stages in the last pass
  static constexpr int N = 1 << LOGN;
  static constexpr int T = 1 << (LOGN - 5);  // threads per workgroup
  static constexpr int E = 32;               // coefficients per thread
  static constexpr int GC = 1 << LC;         // points per pass-C group
  static constexpr int NGC = 32 >> LC;       // pass-C groups per thread
  static constexpr int LDS_WORDS = 33 * T;   // 32-bit words (AB layout is padded 33/32)
  // twiddle table layout (forward and inverse tables have the same shape):
  //   [0,31)                       pass A : index 2^s-1+k               (uniform)
  //   [TWB, TWB+31*32)             pass B : (2^s'-1+k)*32 + hi'
  //   [TWC, TWC+(GC-1)*1024)       pass C : (2^s'-1+k)*1024 + u
  static constexpr int TWB = 32;
  static constexpr int TWC = TWB + 31 * 32;
  static constexpr int TW_TOTAL = TWC + (GC - 1) * 1024;
};

// ---- original (bit-reversed psi table) twiddle index for each slot ----
// psi_rev[idx] = psi^{brev_LOGN(idx)}, 1 <= idx < N.  Stage s (m = 2^s groups)
// butterfly on position p uses idx = 2^s + (p >> (LOGN - s)).
template <int LOGN>
HXD unsigned twA_src(int s, int k) { return (1u << s) + (unsigned)k; }
template <int LOGN>
HXD unsigned twB_src(int sp, int k, unsigned hip)
{
  return (1u << (5 + sp)) + (brev5(hip) << sp) + (unsigned)k;
}
template <int LOGN>
HXD unsigned twC_src(int sp, int k, unsigned u)
{
  return (1u << (10 + sp)) + (brev10(u) << sp) + (unsigned)k;
}

// ---------------------------------------------------------------------
// butterflies
// ---------------------------------------------------------------------
HXD void ct_bfly(uint64_t& X, uint64_t& Y, TW t, uint64_t q, uint64_t q2)
{
  uint64_t x = X;
  x = (x >= q2) ? x - q2 : x;           // [0,2q)
  uint64_t v = shoup_lazy(Y, t, q);     // [0,2q)
  X = x + v;                            // [0,4q)
  Y = x - v + q2;                       // (0,4q)
}
HXD void gs_bfly(uint64_t& X, uint64_t& Y, TW t, uint64_t q, uint64_t q2)
{
  uint64_t x = X, y = Y;                // [0,2q)
  uint64_t s = x + y;
  s = (s >= q2) ? s - q2 : s;           // [0,2q)
  X = s;
  Y = shoup_lazy(x - y + q2, t, q);     // [0,2q)
}
// last inverse stage with N^-1 folded in: X = (x+y)*Ninv, Y = (x-y)*S0*Ninv
HXD void gs_bfly_last(uint64_t& X, uint64_t& Y, TW tNinv, TW tS0Ninv, uint64_t q, uint64_t q2)
{
  uint64_t x = X, y = Y;                // [0,2q)
  X = shoup_lazy(x + y, tNinv, q);      // [0,2q)
  Y = shoup_lazy(x - y + q2, tS0Ninv, q);
}

// ---- bound-tracked butterflies of the row kernels (bounds in units of q) ----
// forward (Cooley-Tukey):  X' = x + T, Y' = x + 4q - T with T = shoup4(Y) in [0,4q), so every
// value grows by 4q per stage; x is brought from [0,16q) to [0,8q) first (CORR) exactly in the
// stages where the bound would otherwise pass 16q.
template <bool CORR>
HXD void ct_bfly4(uint64_t& X, uint64_t& Y, TW t, const QC& c)
{
  uint64_t x = X;
  if constexpr (CORR)
    x = csub(x, c.q8);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HX_SHOUP4_OLD)
  // X' rides on the multiply-add chain; Y' = x + 4q - T = (2x + 4q) - X' (one v_lshl_add_u64 and
  // one 64-bit subtraction; the intermediate may wrap, the value x + 4q - T < 16q does not)
  uint64_t xn = shoup4_acc(Y, t, c.nq, x);
  HX_KEEP64(xn);  // (opaque: otherwise the subtraction below is distributed over xn's halves)
  X = xn;
#ifdef HX_YNOT
  // experiment (tools/ubench/bfly_bench.hip): the 64-bit subtraction as complement-and-add -- two v_not_b32
  // and a v_lshl_add_u64 instead of v_sub_co / v_subb_co with their VCC wait states
  uint64_t nx = ~xn;
  HX_KEEP64(nx);
  uint64_t x2 = (x << 1) + (c.q4 + 1);
  HX_KEEP64(x2);
  Y = x2 + nx;
#else
  uint64_t x2 = (x << 1) + c.q4;
  HX_KEEP64(x2);
  Y = x2 - xn;
#endif
#else
  const uint64_t v = shoup4(Y, t, c.nq);
  X = x + v;
  Y = x + c.q4 - v;
#endif
}
constexpr int fwd_bound_in(int bin, int sp)  // bound of the values entering stage sp of a pass
{
  int b = bin;
  for (int s = 0; s < sp; s++) {
    if (b + 4 > 16)
      b = 8;
    b += 4;
  }
  return b;
}
constexpr bool fwd_corr(int bin, int sp) { return fwd_bound_in(bin, sp) + 4 > 16; }

// inverse (Gentleman-Sande):  X' = x + y, Y' = shoup4(x - y + B q) in [0,4q).  The bound of an
// element after the ex-th executed stage of a pass (pair distance 2^ex) depends only on its
// local index e: 4 if it left that stage on the multiplied arm (bit ex of e set), else twice
// its previous bound, brought back to 8 by one conditional subtraction when that passes 8.
// Both members of a pair have the same previous bound (they differ in bit ex only).
constexpr int inv_bound_after(int bin, int ex, int e)
{
  if (ex < 0)
    return bin;
  if ((e >> ex) & 1)
    return 4;
  const int b = 2 * inv_bound_after(bin, ex - 1, e);
  return b > 8 ? 8 : b;
}
template <int BPREV>  // common bound of x and y
HXD void gs_bfly4(uint64_t& X, uint64_t& Y, TW t, const QC& c)
{
  static_assert(BPREV >= 1 && BPREV <= 8, "bound");
  const uint64_t x = X, y = Y;
  HX_BOUND(x, BPREV, c.q);
  HX_BOUND(y, BPREV, c.q);
  uint64_t s = x + y;
  if constexpr (2 * BPREV > 8)
    s = csub(s, c.q8);
  X = s;
  Y = shoup4(x + c.q8 - y, t, c.nq);  // one offset constant (8q >= any bound here) keeps q4 dead
}
// last inverse stage with N^-1 folded in: X = (x+y)*Ninv, Y = (x-y)*S0*Ninv, both in [0,4q)
template <int BPREV>
HXD void gs_bfly4_last(uint64_t& X, uint64_t& Y, TW tNinv, TW tS0Ninv, const QC& c)
{
  static_assert(BPREV >= 1 && BPREV <= 8, "bound");
  const uint64_t x = X, y = Y;
  HX_BOUND(x, BPREV, c.q);
  HX_BOUND(y, BPREV, c.q);
  X = shoup4(x + y, tNinv, c.nq);
  Y = shoup4(x + c.q8 - y, tS0Ninv, c.nq);
}

// ---- Proth-form butterflies (mont_acc above).  Bounds in SIXTEENTHS of q. ----
// forward:  X' = x + R,  Y' = x + 2q - R = (2x + 2q) - X'  (0 < R < 2q; the offset must be a multiple of q).
// Both outputs are below (Bx + 2) q; the multiplied operand must be below 207/16 q and x + 2q below 16q, so
// BOTH arms are taken below 8q first (one conditional subtraction each) in the stages where the running bound
// has passed 207: three of the fourteen stages at N = 2^14 for canonical input -- the same number of
// conditional subtractions as the Shoup butterfly (six stages, one arm).
constexpr int P16_YMAX = 207;  // multiplied operand < 12.9375 q  (mont_acc: no 64-bit sum wraps)
constexpr int p16_corr(int b) { return b > P16_YMAX ? 128 : b; }
constexpr int p16_k(int) { return 32; }
constexpr int p16_fwd_in(int bin16, int sp)  // bound of the values entering stage sp of a pass
{
  int b = bin16;
  for (int s = 0; s < sp; s++)
    b = p16_corr(b) + p16_k(b);
  return b;
}
#if defined(HX_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#define HX_BOUND16(x, B16, q)                                                            \
  do {                                                                                   \
    if ((unsigned __int128)(x) * 16 >= (unsigned __int128)(B16) * (q)) {                 \
      std::fprintf(stderr, "bound violated at %s:%d (B16=%d)\n", __FILE__, __LINE__, (int)(B16)); \
      std::abort();                                                                      \
    }                                                                                    \
  } while (0)
#else
#define HX_BOUND16(x, B16, q) ((void)0)
#endif
template <int B16>  // common bound of x and y on entry
HXD void ct_bfly_p(uint64_t& X, uint64_t& Y, TWM W, const QC& c)
{
  static_assert(B16 >= 1 && B16 <= 256, "bound");
  uint64_t x = X, y = Y;
  HX_BOUND16(x, B16, c.q);
  HX_BOUND16(y, B16, c.q);
  if constexpr (B16 > P16_YMAX) {
    x = csub(x, c.q8);
    y = csub(y, c.q8);
  }
  static_assert(p16_corr(B16) + p16_k(B16) <= 256, "x + K must stay below 16q");
  uint64_t xn = mont_acc(y, W, c, x);
  HX_KEEP64(xn);
  X = xn;
#if defined(__HIP_DEVICE_COMPILE__)
  // (2x + 2q as ONE v_lshl_add_u64: written out, because the compiler knows q2 = q << 1 and prefers (x + q) << 1)
  uint64_t x2;
  asm("v_lshl_add_u64 %0, %1, 1, %2" : "=v"(x2) : "v"(x), "s"(c.q2));
#else
  uint64_t x2 = (x << 1) + c.q2;
#endif
  Y = x2 - xn;
}
// inverse:  X' = x + y,  Y' = (x + 4q - y) W  in (0, 2q).  Every value entering a butterfly is below 4q, so
// the multiplied operand is below 8q (R < 1.5q + ...: bound 2); the sum is taken back below 4q when its
// operands' common bound has passed 2 -- the same places as the Shoup butterfly's, at half its bounds.
constexpr int inv_bound_after_p(int bin, int ex, int e)
{
  if (ex < 0)
    return bin;
  if ((e >> ex) & 1)
    return 2;
  const int b = 2 * inv_bound_after_p(bin, ex - 1, e);
  return b > 4 ? 4 : b;
}
template <int BPREV>
HXD void gs_bfly_p(uint64_t& X, uint64_t& Y, TWM W, const QC& c)
{
  static_assert(BPREV >= 1 && BPREV <= 4, "bound");
  const uint64_t x = X, y = Y;
  HX_BOUND(x, BPREV, c.q);
  HX_BOUND(y, BPREV, c.q);
  uint64_t s = x + y;
  if constexpr (2 * BPREV > 4)
    s = csub(s, c.q4);
  X = s;
  Y = mont_mul(x + c.q4 - y, W, c);
}
template <int BPREV>
HXD void gs_bfly_p_last(uint64_t& X, uint64_t& Y, TWM wNinv, TWM wS0Ninv, const QC& c)
{
  static_assert(BPREV >= 1 && BPREV <= 4, "bound");
  const uint64_t x = X, y = Y;
  HX_BOUND(x, BPREV, c.q);
  HX_BOUND(y, BPREV, c.q);
  X = mont_mul(x + y, wNinv, c);
  Y = mont_mul(x + c.q4 - y, wS0Ninv, c);
}

// ---- the two arithmetics of the row kernels as policies of run_pass / RowNTT ----
// U = bound units per q (1: whole q; 16: sixteenths).  fwd_out/inv_after: the compile-time schedules.
struct ArShoup {
  using Tw = TW;
  static constexpr bool PROTH = false;
  static constexpr int U = 1;
  static constexpr int FWD_LOAD_MAX = 12;   // largest IO::LOAD_BOUND (units of q) a forward pass accepts
  static constexpr int INV_OUT = 4;         // bound (units of q) of what the inverse transform hands its store
  static constexpr int fwd_in(int bin, int sp) { return fwd_bound_in(bin, sp); }
  static constexpr int inv_after(int bin, int ex, int e) { return inv_bound_after(bin, ex, e); }
  static constexpr int inv_cap() { return 8; }
  template <int BIN, int SP>
  static HXD void ct(uint64_t& X, uint64_t& Y, Tw t, const QC& c)
  {
    HX_BOUND(X, fwd_bound_in(BIN, SP), c.q);
    ct_bfly4<fwd_corr(BIN, SP)>(X, Y, t, c);
  }
  template <int BPREV>
  static HXD void gs(uint64_t& X, uint64_t& Y, Tw t, const QC& c) { gs_bfly4<BPREV>(X, Y, t, c); }
  template <int BPREV>
  static HXD void gs_last(uint64_t& X, uint64_t& Y, Tw tN, Tw tS, const QC& c) { gs_bfly4_last<BPREV>(X, Y, tN, tS, c); }
};
struct ArProth {
  using Tw = TWM;
  static constexpr bool PROTH = true;
  static constexpr int U = 16;
  static constexpr int FWD_LOAD_MAX = 12;
  static constexpr int INV_OUT = 2;
  static constexpr int fwd_in(int bin16, int sp) { return p16_fwd_in(bin16, sp); }
  static constexpr int inv_after(int bin, int ex, int e) { return inv_bound_after_p(bin, ex, e); }
  static constexpr int inv_cap() { return 4; }
  template <int BIN, int SP>
  static HXD void ct(uint64_t& X, uint64_t& Y, Tw t, const QC& c) { ct_bfly_p<p16_fwd_in(BIN, SP)>(X, Y, t, c); }
  template <int BPREV>
  static HXD void gs(uint64_t& X, uint64_t& Y, Tw t, const QC& c) { gs_bfly_p<BPREV>(X, Y, t, c); }
  template <int BPREV>
  static HXD void gs_last(uint64_t& X, uint64_t& Y, Tw tN, Tw tS, const QC& c) { gs_bfly_p_last<BPREV>(X, Y, tN, tS, c); }
};

// ---------------------------------------------------------------------
// register passes.  v[32] is the thread's coefficient file.
//
// A pass is REP independent radix-2^S sub-transforms on v[rep*2^S ..]; stage sp
// pairs element e with e + (2^(S-1) >> sp) and all butterflies of "group"
// (sp, k = e >> (S - sp)) share one twiddle.  Groups are visited in a fixed flat
// order (forward: sp ascending, inverse: sp descending) and their twiddles are
// fetched through a PF-deep software queue: the load for group i+PF is issued
// right after group i has been computed and is made data-dependent on one of its
// results (HX_LAUNDER), so hipcc can neither hoist all 31 loads of a pass to its
// top (124 VGPRs, costs a wave of occupancy) nor sink them to the point of use.
// ---------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HX_NO_LAUNDER)
#define HX_LAUNDER(ptr, dep) asm volatile("" : "+v"(ptr) : "v"(dep))
#else
#define HX_LAUNDER(ptr, dep) ((void)(dep))
#endif
// Round 1: depth 2 (with the per-phase work-item id and the IO fences below) was what kept the
// N = 2^14 kernels at or near zero scratch inside the 128-VGPR budget of two workgroups per CU.
// Round 2: with the LDS reads no longer volatile the kernels sit near 100 VGPRs and the queue can be
// deeper -- forward 4, inverse 6 (measured: forward 0.546 -> 0.510 ms per 6400 rows, inverse
// 0.631 -> 0.593); depth 6 in the forward kernels spills in the N = 2^15 mod-down apply, depth 8
// spills in several (profiles/r02_variants_twiddle_queue.txt).
#ifndef HX_TW_PF
#define HX_TW_PF 4
#endif
#ifndef HX_TW_PF_INV
#define HX_TW_PF_INV 6
#endif
// the Proth-form rows' queue: an entry is 8 bytes (two VGPRs) instead of 16, so the same registers hold a queue twice
// as deep (variants: tools/build_variant.sh NAME -DHX_TW_PF_P=8 -DHX_TW_PF_INV_P=8)
#ifndef HX_TW_PF_P
#define HX_TW_PF_P HX_TW_PF
#endif
#ifndef HX_TW_PF_INV_P
#define HX_TW_PF_INV_P HX_TW_PF_INV
#endif
#ifndef HX_IO_GROUP
#define HX_IO_GROUP 4
#endif
#ifndef HX_BF_FENCE
#define HX_BF_FENCE 4
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define HX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define HX_SCHED_FENCE() ((void)0)
#endif
// The fused IO functors (mod-down prep/apply) do tens of instructions per element; without a
// fence the scheduler interleaves many elements' temporaries while all 32 coefficients are still
// live.  HX_IO_FENCE(i) stops code motion across every HX_IO_GROUP-th element.
#if defined(__HIP_DEVICE_COMPILE__) && HX_IO_GROUP > 0
#define HX_IO_FENCE(i)                          \
  do {                                          \
    if (((i) % HX_IO_GROUP) == HX_IO_GROUP - 1) \
      __builtin_amdgcn_sched_barrier(0);        \
  } while (0)
#else
#define HX_IO_FENCE(i) ((void)0)
#endif

// Twiddle-table access, overloaded on the table handle: a plain pointer here (CPU replay, and
// valid on the device), a buffer resource in ntt_kernels.hip (one address VGPR per load instead
// of a 64-bit pointer).  tw_uni: wave-uniform entry; tw_vec: entry base + lane + off, whose
// issue is tied to `dep` (see HX_LAUNDER above run_pass).
HXD TW tw_uni(const TW* tw, unsigned i) { return tw[i]; }
HXD TW tw_vec(const TW* tw, unsigned base, unsigned lane, unsigned off, uint32_t dep)
{
  HX_LAUNDER(off, dep);
  return tw[base + lane + off];
}
// the Proth-form tables: same positions, 8-byte entries W = w 2^64 mod q
HXD TWM tw_uni(const TWM* tw, unsigned i) { return tw[i]; }
HXD TWM tw_vec(const TWM* tw, unsigned base, unsigned lane, unsigned off, uint32_t dep)
{
  HX_LAUNDER(off, dep);
  return tw[base + lane + off];
}

template <int S, bool INV>
HXD constexpr int grp_sp(int i)
{
  int sp = INV ? S - 1 : 0;
  while (i >= (1 << sp)) {
    i -= (1 << sp);
    sp += INV ? -1 : 1;
  }
  return sp;
}
template <int S, bool INV>
HXD constexpr int grp_k(int i)
{
  int sp = INV ? S - 1 : 0;
  while (i >= (1 << sp)) {
    i -= (1 << sp);
    sp += INV ? -1 : 1;
  }
  return i;
}

// compile-time loop: f(integral_constant<int, I>) for I in [0, N)
template <int I>
struct IC {
  static constexpr int value = I;
};
template <int I, int N, class F>
HXD void static_for(F&& f)
{
  if constexpr (I < N) {
    f(IC<I>{});
    static_for<I + 1, N>(f);
  }
}

// NGRUN: number of leading groups (flat order) to run per repetition; the inverse
// pass A runs 2^S-2 groups here and finishes stage 0 with gs_bfly4_last.
// BIN: bound (units of q) of every value entering the pass; pass_bound_out gives the bound
// of every value leaving it (forward: exact schedule above; inverse: 8, or less for short runs).
template <class AR, int S, bool INV, int NGRUN, int BIN>
constexpr int pass_bound_out()
{
  if constexpr (!INV) {
    static_assert(NGRUN == (1 << S) - 1, "forward passes run all stages");
    return AR::fwd_in(BIN, S);
  } else {
    // executed stages: ex = 0 .. nst-1 (whole stages only)
    int nst = 0, g = NGRUN;
    for (int sp = S - 1; sp >= 0 && g >= (1 << sp); sp--) {
      g -= (1 << sp);
      nst++;
    }
    int b = 0;
    for (int e = 0; e < (1 << S); e++) {
      const int be = AR::inv_after(BIN, nst - 1, e);
      b = be > b ? be : b;
    }
    return b;
  }
}
template <class AR, int S, bool INV, int REP, int NGRUN, int BIN, class Fetch>
HXD void run_pass(uint64_t (&v)[32], const QC& c, Fetch fetch)
{
  constexpr int PF = AR::PROTH ? (INV ? HX_TW_PF_INV_P : HX_TW_PF_P) : (INV ? HX_TW_PF_INV : HX_TW_PF);
  constexpr int TOT = REP * NGRUN;
  typename AR::Tw tq[PF];
  static_for<0, (PF < TOT ? PF : TOT)>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int rep = i / NGRUN, sp = grp_sp<S, INV>(i % NGRUN), k = grp_k<S, INV>(i % NGRUN);
    tq[i] = fetch(rep, sp, k, (uint32_t)v[0]);
  });
  static_for<0, TOT>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int rep = i / NGRUN, ii = i % NGRUN;
    constexpr int sp = grp_sp<S, INV>(ii), k = grp_k<S, INV>(ii);
    constexpr int half = (1 << (S - 1)) >> sp;
    constexpr int base = rep * (1 << S) + k * 2 * half;
    const typename AR::Tw t = tq[i % PF];
    static_for<0, half>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (INV) {
        constexpr int ex = S - 1 - sp;  // executed-stage index, pair distance 2^ex
        constexpr int bprev = AR::inv_after(BIN, ex - 1, k * 2 * half + j);
        AR::template gs<bprev>(v[base + j], v[base + j + half], t, c);
      } else {
        AR::template ct<BIN, sp>(v[base + j], v[base + j + half], t, c);
      }
    });
    if constexpr (i + PF < TOT) {
      constexpr int n = i + PF;
      constexpr int nrep = n / NGRUN, nsp = grp_sp<S, INV>(n % NGRUN), nk = grp_k<S, INV>(n % NGRUN);
      tq[i % PF] = fetch(nrep, nsp, nk, (uint32_t)v[base]);
    }
    // keep the instruction scheduler from interleaving more than HX_BF_FENCE butterflies'
    // worth of independent groups (the single-butterfly groups of the fine stages otherwise
    // pile up a dozen twiddle loads and partial products, and the register file spills)
    if constexpr (HX_BF_FENCE > 0) {
      if constexpr (half >= HX_BF_FENCE)
        HX_SCHED_FENCE();
      else if constexpr ((k + 1) % (HX_BF_FENCE / half) == 0)
        HX_SCHED_FENCE();
    }
  });
}
// (the pre-round-5 spelling: the Shoup arithmetic)
template <int S, bool INV, int REP, int NGRUN, int BIN, class Fetch>
HXD void run_pass(uint64_t (&v)[32], const QC& c, Fetch fetch)
{
  run_pass<ArShoup, S, INV, REP, NGRUN, BIN>(v, c, fetch);
}

// ---------------------------------------------------------------------
// LDS transposes, one 32-bit half at a time (half = 0: low words, 1: high).
// Position p of the row (LOGN bits):
//   pass A thread: tid = p & (T-1),          e  = p >> (LOGN-5)
//   pass B thread: tid = lo*32 + brev5(hi),  eB = (p >> LC) & 31,
//                  hi = p >> (LOGN-5), lo = p & (GC-1)
//   pass C thread: u = brev10(p >> LC) = tid + T*gi, e' = p & (GC-1)
// AB layout: word = brev5(hi) + 33*(p & (T-1))    (33: conflict-free both ways)
// BC layout: word = u + 1024*e'
// ---------------------------------------------------------------------
template <int LOGN>
HXD unsigned ab_addr_A(unsigned tid, int e) { return brev5((unsigned)e) + 33u * tid; }
template <int LOGN>
HXD unsigned ab_addr_B(unsigned tid, int eB)
{
  constexpr int LC = Geo<LOGN>::LC;
  return (tid & 31u) + 33u * (((unsigned)eB << LC) + (tid >> 5));
}
template <int LOGN>
HXD unsigned bc_addr_B(unsigned tid, int eB)
{
  return (tid & 31u) + 32u * brev5((unsigned)eB) + 1024u * (tid >> 5);
}
template <int LOGN>
HXD unsigned bc_addr_C(unsigned tid, int i)  // i = gi*GC + e'
{
  constexpr int LC = Geo<LOGN>::LC;
  constexpr int T = Geo<LOGN>::T;
  unsigned gi = (unsigned)i >> LC, ep = (unsigned)i & ((1u << LC) - 1u);
  return (tid + (unsigned)T * gi) + 1024u * ep;
}

// LDS reads of the two transposes.  Each of the 32 words a thread reads goes into its own
// register (the low or high half of one 64-bit coefficient); a compiler-fused ds_read2 would
// deliver halves of two DIFFERENT coefficients in one register pair and un-pairing them costs a
// v_mov per word plus the registers to hold both copies (the inverse kernels spilled on exactly
// that).  Round 1 prevented the fusion with a `volatile` read -- which the compiler cannot prove
// to be an LDS access (address-space inference skips volatile), so it became a system-scope
// flat_load_dword followed by s_waitcnt vmcnt(0), 128 fully serialised round trips per thread.
// Now: 32 explicit ds_read_b32 (one base VGPR per 64 KiB of LDS, the per-element part in the
// 16-bit immediate offset) in flight together, one s_waitcnt.  `Cst(e)` = word offset of element
// e relative to element 0's word `base`, a constexpr function of e only.
template <class Cst, int E>
struct LdsOff {
  static constexpr unsigned bytes = Cst::at(E) * 4u;
  static constexpr unsigned chunk = bytes >> 16, imm = bytes & 0xffffu;
};
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HX_LDS_RD_PLAIN)
template <int OFF>
__device__ __forceinline__ uint32_t ds_rd_b32(unsigned a)
{
  uint32_t x;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x) : "v"(a), "n"(OFF) : "memory");
  return x;
}
template <class Cst>
HXD void lds_read32(const uint32_t* lds, unsigned base, uint32_t (&o)[32])
{
  // low 32 bits of a generic (flat) pointer into the LDS aperture = the LDS byte address
  const unsigned a0 = (unsigned)(uintptr_t)lds + base * 4u;
  const unsigned a1 = a0 + 0x10000u, a2 = a0 + 0x20000u;
  static_for<0, 32>([&](auto I) {
    constexpr int e = decltype(I)::value;
    using O = LdsOff<Cst, e>;
    static_assert(O::chunk <= 2, "LDS offset");
    o[e] = ds_rd_b32<(int)O::imm>(O::chunk == 0 ? a0 : (O::chunk == 1 ? a1 : a2));
  });
  // the results exist only after the wait: every consumer is made to depend on it (two asm
  // statements because one takes at most 30 operands; volatile asms keep their order)
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]),
                 "+v"(o[7]), "+v"(o[8]), "+v"(o[9]), "+v"(o[10]), "+v"(o[11]), "+v"(o[12]),
                 "+v"(o[13]), "+v"(o[14]), "+v"(o[15])
               :
               : "memory");
  asm volatile(""
               : "+v"(o[16]), "+v"(o[17]), "+v"(o[18]), "+v"(o[19]), "+v"(o[20]), "+v"(o[21]),
                 "+v"(o[22]), "+v"(o[23]), "+v"(o[24]), "+v"(o[25]), "+v"(o[26]), "+v"(o[27]),
                 "+v"(o[28]), "+v"(o[29]), "+v"(o[30]), "+v"(o[31])
               :
               : "memory");
}
#else
template <class Cst>
HXD void lds_read32(const uint32_t* lds, unsigned base, uint32_t (&o)[32])
{
  for (int e = 0; e < 32; e++)
    o[e] = lds[base + Cst::at(e)];
}
#endif
// element-e word offsets of the four read patterns relative to element 0 (cf. the *_addr_*
// functions above: each is base(tid) + a function of e alone)
template <int LOGN>
struct CstAbB {  // ab_addr_B(tid, e) - ab_addr_B(tid, 0)
  static constexpr unsigned at(int e) { return 33u * ((unsigned)e << Geo<LOGN>::LC); }
};
template <int LOGN>
struct CstBcC {  // bc_addr_C(tid, i) - bc_addr_C(tid, 0)
  static constexpr unsigned at(int i)
  {
    return (unsigned)Geo<LOGN>::T * ((unsigned)i >> Geo<LOGN>::LC) + 1024u * ((unsigned)i & ((1u << Geo<LOGN>::LC) - 1u));
  }
};
template <int LOGN>
struct CstBcB {  // bc_addr_B(tid, e) - bc_addr_B(tid, 0)
  static constexpr unsigned at(int e) { return 32u * brev_bits((unsigned)e, 5); }
};
template <int LOGN>
struct CstAbA {  // ab_addr_A(tid, e) - ab_addr_A(tid, 0)
  static constexpr unsigned at(int e) { return brev_bits((unsigned)e, 5); }
};
HXD uint32_t half_of(uint64_t x, int half) { return half ? (uint32_t)(x >> 32) : (uint32_t)x; }
HXD void set_half(uint64_t& x, int half, uint32_t w)
{
  x = half ? ((x & 0xffffffffull) | ((uint64_t)w << 32)) : ((x & 0xffffffff00000000ull) | w);
}

// global <-> register index maps
// natural coefficient index held in v[e] of pass-A thread tid
template <int LOGN>
HXD unsigned coef_index(unsigned tid, int e) { return ((unsigned)e << (LOGN - 5)) + tid; }
// natural evaluation index j (= 2j+1-th power) held in v[i] of pass-C thread tid
template <int LOGN>
HXD unsigned eval_index(unsigned tid, int i)
{
  constexpr int LC = Geo<LOGN>::LC;
  constexpr int T = Geo<LOGN>::T;
  unsigned gi = (unsigned)i >> LC, ep = (unsigned)i & ((1u << LC) - 1u);
  return brev_bits(ep, LC) * 1024u + tid + (unsigned)T * gi;
}

// constant (thread-independent) part of the two index maps: index = tid + const
template <int LOGN>
HXD unsigned coef_const(int e) { return (unsigned)e << (LOGN - 5); }
template <int LOGN>
HXD unsigned eval_const(int i)
{
  constexpr int LC = Geo<LOGN>::LC;
  constexpr int T = Geo<LOGN>::T;
  unsigned gi = (unsigned)i >> LC, ep = (unsigned)i & ((1u << LC) - 1u);
  return brev_bits(ep, LC) * 1024u + (unsigned)T * gi;
}

// An IO functor may state SKIP_LOAD = true: the inverse transform then starts from what the register
// file already holds (the convolution kernel: forward transform, pointwise product, inverse transform
// without leaving the registers).
template <class IO, class = void>
struct io_skip_load : std::false_type {};
template <class IO>
struct io_skip_load<IO, std::void_t<decltype(IO::SKIP_LOAD)>> : std::integral_constant<bool, IO::SKIP_LOAD> {};

// The bound (units of q) of what an IO functor's load hands the forward transform: IO::LOAD_BOUND, or -- when the
// functor's own arithmetic depends on the row's arithmetic (the mod-down apply loads: a Shoup product is below 4q, a
// Proth-form one below 2q) -- IO::load_bound<AR>().
template <class IO, class AR, class = void>
struct io_load_bound : std::integral_constant<int, IO::LOAD_BOUND> {};
template <class IO, class AR>
struct io_load_bound<IO, AR, std::void_t<decltype(IO::template load_bound<AR>())>>
    : std::integral_constant<int, IO::template load_bound<AR>()> {};

// ... and of what the inverse transform starts from (1: canonical; the convolution kernel's Proth-form pointwise
// product hands over values below 2q: IO::inv_load_bound<AR>())
template <class IO, class AR, class = void>
struct io_inv_load_bound : std::integral_constant<int, 1> {};
template <class IO, class AR>
struct io_inv_load_bound<IO, AR, std::void_t<decltype(IO::template inv_load_bound<AR>())>>
    : std::integral_constant<int, IO::template inv_load_bound<AR>()> {};

// Plain-pointer row accessor (CPU replay; also valid on the device).
struct PtrIO {
  static constexpr int LOAD_BOUND = 1;
  static constexpr bool LAZY_STORE = false;
  static constexpr bool PIPELINED = false;
  struct StorePrefetch {};
  const uint64_t* in;
  uint64_t* out;
  HXD uint64_t load(unsigned tid, unsigned c) const { return in[tid + c]; }
  HXD void store(unsigned tid, unsigned c, uint64_t v) const { out[tid + c] = v; }
  HXD TW last_tw(TW def, int) const { return def; }
  HXD TWM last_tw(TWM def, int) const { return def; }
};

HXD uint64_t norm4(uint64_t x, uint64_t q, uint64_t q2)  // [0,4q) -> [0,q)
{
  x = (x >= q2) ? x - q2 : x;
  return (x >= q) ? x - q : x;
}
HXD uint64_t norm2(uint64_t x, uint64_t q)  // [0,2q) -> [0,q)
{
  return (x >= q) ? x - q : x;
}
// [0, B q) -> [0, q) for a compile-time B <= 16.
// EST (requires c.mu32 != 0, i.e. q > 2^32): quotient estimate from the 32-bit reciprocal,
// e = floor(x mu32 / 2^64) is floor(x/q) or one less (x mu32 < 2^96 is formed exactly below;
// 0 <= x/q - x mu32/2^64 < x/2^64 <= 1), so x - e q is in [0, 2q): two multiplications and ONE
// conditional subtraction instead of four.  Callers branch on c.mu32 once per phase, outside
// their element loops (a branch per element fragments the schedule and spills).
template <int B, bool EST = false>
HXD uint64_t norm_from(uint64_t x, const QC& c)
{
  static_assert(B >= 1 && B <= 16, "bound");
  HX_BOUND(x, B, c.q);
  if constexpr (EST && B > 4) {
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    const uint64_t t = (uint64_t)xh * c.mu32 + mulhi32(xl, c.mu32);
    const uint32_t e = (uint32_t)(t >> 32);
    const uint32_t nl = (uint32_t)c.nq, nh = (uint32_t)(c.nq >> 32);
    uint64_t r = (uint64_t)e * nl + x;  // x + e (2^64 - q) mod 2^64
    r += (uint64_t)(e * nh) << 32;
    HX_BOUND(r, 2, c.q);
    return csub(r, c.q);
  } else {
    if constexpr (B > 8)
      x = csub(x, c.q8);
    if constexpr (B > 4)
      x = csub(x, c.q4);
    if constexpr (B > 2)
      x = csub(x, c.q + c.q);
    if constexpr (B > 1)
      x = csub(x, c.q);
    return x;
  }
}

// ---------------------------------------------------------------------
// The transform as barrier-separated phases.  The HIP kernel runs
//   phase<0>; barrier; phase<1>; barrier; ... phase<7>
// for its own tid; the CPU replay in tests/ runs every tid through phase k
// before moving to phase k+1.  v = 32 coefficients, nl = 32 staged low words.
// ---------------------------------------------------------------------
template <int LOGN, class AR = ArShoup>
struct RowNTT {
  using G = Geo<LOGN>;
  static constexpr int NPHASE = 8;


  // -------- forward: coefficients (natural) -> evaluations (natural) -----
  template <int PH, class IO, class TWS>
  static HXD void fwd(unsigned tid, uint64_t (&v)[32], uint32_t (&nl)[32], uint32_t* lds,
                      const IO& io, const TWS& tw, const QC& c)
  {
    // bounds between the forward passes; the IO functor states the bound of what it loads
    // (IO::LOAD_BOUND, 1 = canonical) and whether its store takes the lazy value (IO::LAZY_STORE)
    // (in the policy's units, AR::U per q; the IO functors state and take bounds in whole q)
    constexpr int LB = io_load_bound<IO, AR>::value;
    constexpr int L0 = LB * AR::U;
    constexpr int FA = pass_bound_out<AR, 5, false, 31, L0>();
    constexpr int FB = pass_bound_out<AR, 5, false, 31, FA>();
    constexpr int FCU = pass_bound_out<AR, G::LC, false, G::GC - 1, FB>();
    constexpr int FC = (FCU + AR::U - 1) / AR::U;
    static_assert(LB <= AR::FWD_LOAD_MAX && FA <= 16 * AR::U && FB <= 16 * AR::U && FCU <= 16 * AR::U, "lazy bounds");
    if constexpr (PH == 0) {
      if constexpr (IO::PIPELINED) {
        io.template load_all<LOGN, AR>(tid, v, c);  // (software-pipelined element loads, see ModDownIO)
      } else {
#pragma unroll
        for (int e = 0; e < 32; e++) {
          v[e] = io.load(tid, coef_const<LOGN>(e));
          HX_IO_FENCE(e);
        }
      }
      run_pass<AR, 5, false, 1, 31, L0>(v, c, [&](int, int sp, int k, uint32_t) {
        return tw_uni(tw, (unsigned)((1 << sp) - 1 + k));  // uniform: scalar loads
      });
#pragma unroll
      for (int e = 0; e < 32; e++)
        lds[ab_addr_A<LOGN>(tid, e)] = half_of(v[e], 0);
    } else if constexpr (PH == 1) {
      lds_read32<CstAbB<LOGN>>(lds, ab_addr_B<LOGN>(tid, 0), nl);
    } else if constexpr (PH == 2) {
#pragma unroll
      for (int e = 0; e < 32; e++)
        lds[ab_addr_A<LOGN>(tid, e)] = half_of(v[e], 1);
    } else if constexpr (PH == 3) {
      uint32_t nh[32];
      lds_read32<CstAbB<LOGN>>(lds, ab_addr_B<LOGN>(tid, 0), nh);
#pragma unroll
      for (int e = 0; e < 32; e++)
        v[e] = ((uint64_t)nh[e] << 32) | nl[e];
      run_pass<AR, 5, false, 1, 31, FA>(v, c, [&](int, int sp, int k, uint32_t dep) {
        return tw_vec(tw, G::TWB, tid & 31u, ((1u << sp) - 1u + (unsigned)k) * 32u, dep);
      });
    } else if constexpr (PH == 4) {
#pragma unroll
      for (int e = 0; e < 32; e++)
        lds[bc_addr_B<LOGN>(tid, e)] = half_of(v[e], 0);
    } else if constexpr (PH == 5) {
      lds_read32<CstBcC<LOGN>>(lds, bc_addr_C<LOGN>(tid, 0), nl);
    } else if constexpr (PH == 6) {
#pragma unroll
      for (int e = 0; e < 32; e++)
        lds[bc_addr_B<LOGN>(tid, e)] = half_of(v[e], 1);
    } else if constexpr (PH == 7) {
      uint32_t nh[32];
      lds_read32<CstBcC<LOGN>>(lds, bc_addr_C<LOGN>(tid, 0), nh);
#pragma unroll
      for (int i = 0; i < 32; i++)
        v[i] = ((uint64_t)nh[i] << 32) | nl[i];
      // a fused store that reads its own operand row starts on it before the last register pass
      typename IO::StorePrefetch pre;
      if constexpr (IO::LAZY_STORE)
        io.template store_prefetch<LOGN>(tid, pre);
      run_pass<AR, G::LC, false, G::NGC, G::GC - 1, FB>(v, c, [&](int gi, int sp, int k, uint32_t dep) {
        return tw_vec(tw, G::TWC, tid, ((1u << sp) - 1u + (unsigned)k) * 1024u + (unsigned)(G::T * gi), dep);
      });
      // (uniform branches are taken once here / inside store_all, never per element: a branch
      // per element fragments the schedule into 32 blocks and the register file spills)
      if constexpr (IO::LAZY_STORE) {
        if (c.mu32)
          io.template store_all<LOGN, AR, FC, true>(tid, v, c, pre);
        else
          io.template store_all<LOGN, AR, FC, false>(tid, v, c, pre);
      } else if (c.mu32) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
          io.store(tid, eval_const<LOGN>(i), norm_from<FC, true>(v[i], c));
          HX_IO_FENCE(i);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 32; i++) {
          io.store(tid, eval_const<LOGN>(i), norm_from<FC, false>(v[i], c));
          HX_IO_FENCE(i);
        }
      }
    }
  }

  // -------- inverse: evaluations (natural) -> coefficients (natural) -----
  template <int PH, class IO, class TWS>
  static HXD void inv(unsigned tid, uint64_t (&v)[32], uint32_t (&nl)[32], uint32_t* lds,
                      const IO& io, const TWS& tw, const QC& c)
  {
    // bounds (units of q) of the register file between the inverse passes (inputs canonical unless the IO says otherwise)
    constexpr int I0 = io_inv_load_bound<IO, AR>::value;
    constexpr int IC_ = pass_bound_out<AR, G::LC, true, G::GC - 1, I0>();  // after inverse pass C
    constexpr int IB = pass_bound_out<AR, 5, true, 31, IC_>();
    static_assert(I0 <= AR::inv_cap() && IC_ <= AR::inv_cap() && IB <= AR::inv_cap(), "lazy bounds");
    if constexpr (PH == 0) {
      if constexpr (!io_skip_load<IO>::value) {
#pragma unroll
        for (int i = 0; i < 32; i++)
          v[i] = io.load(tid, eval_const<LOGN>(i));
      }
      run_pass<AR, G::LC, true, G::NGC, G::GC - 1, I0>(v, c, [&](int gi, int sp, int k, uint32_t dep) {
        return tw_vec(tw, G::TWC, tid, ((1u << sp) - 1u + (unsigned)k) * 1024u + (unsigned)(G::T * gi), dep);
      });
#pragma unroll
      for (int i = 0; i < 32; i++)
        lds[bc_addr_C<LOGN>(tid, i)] = half_of(v[i], 0);
    } else if constexpr (PH == 1) {
      lds_read32<CstBcB<LOGN>>(lds, bc_addr_B<LOGN>(tid, 0), nl);
    } else if constexpr (PH == 2) {
#pragma unroll
      for (int i = 0; i < 32; i++)
        lds[bc_addr_C<LOGN>(tid, i)] = half_of(v[i], 1);
    } else if constexpr (PH == 3) {
      uint32_t nh[32];
      lds_read32<CstBcB<LOGN>>(lds, bc_addr_B<LOGN>(tid, 0), nh);
#pragma unroll
      for (int e = 0; e < 32; e++)
        v[e] = ((uint64_t)nh[e] << 32) | nl[e];
      run_pass<AR, 5, true, 1, 31, IC_>(v, c, [&](int, int sp, int k, uint32_t dep) {
        return tw_vec(tw, G::TWB, tid & 31u, ((1u << sp) - 1u + (unsigned)k) * 32u, dep);
      });
    } else if constexpr (PH == 4) {
#pragma unroll
      for (int e = 0; e < 32; e++)
        lds[ab_addr_B<LOGN>(tid, e)] = half_of(v[e], 0);
    } else if constexpr (PH == 5) {
      lds_read32<CstAbA<LOGN>>(lds, ab_addr_A<LOGN>(tid, 0), nl);
    } else if constexpr (PH == 6) {
#pragma unroll
      for (int e = 0; e < 32; e++)
        lds[ab_addr_B<LOGN>(tid, e)] = half_of(v[e], 1);
    } else if constexpr (PH == 7) {
      uint32_t nh[32];
      lds_read32<CstAbA<LOGN>>(lds, ab_addr_A<LOGN>(tid, 0), nh);
#pragma unroll
      for (int e = 0; e < 32; e++)
        v[e] = ((uint64_t)nh[e] << 32) | nl[e];
      // stages 4..1 (30 groups), then stage 0 with N^-1 folded in:
      // slot 0 = S0*N^-1, slot 31 = N^-1
      run_pass<AR, 5, true, 1, 30, IB>(v, c, [&](int, int sp, int k, uint32_t) {
        return tw_uni(tw, (unsigned)((1 << sp) - 1 + k));
      });
      {
        // (the IO functor may substitute its own pair: a constant factor folded into N^-1)
        const typename AR::Tw tS = io.last_tw(tw_uni(tw, 0), 0), tN = io.last_tw(tw_uni(tw, 31), 1);
        static_for<0, 16>([&](auto J) {
          constexpr int j = decltype(J)::value;
          AR::template gs_last<AR::inv_after(IB, 3, j)>(v[j], v[j + 16], tN, tS, c);
          if constexpr (HX_BF_FENCE > 0 && j % 2 == 1)
            HX_SCHED_FENCE();
        });
      }
#pragma unroll
      for (int e = 0; e < 32; e++) {
        io.store(tid, coef_const<LOGN>(e), norm_from<AR::INV_OUT>(v[e], c));
        HX_IO_FENCE(e);
      }
    }
  }
};

// Host-side table builder (used by the library when a prime is registered and
// by the CPU replay test).  psi = primitive 2N-th root of unity (w0).
// mulmod/powmod/invmod are supplied by the caller (exact 128-bit arithmetic).
// Generalised form: the tables of sub-transform `g` of a 2^(LOGN+OUT)-point transform whose
// first OUT stages were done elsewhere (OUT = 0, g = 0: the plain transform).  After OUT
// Cooley-Tukey stages the block with top bits g continues with the same network, its stage-s'
// group-j butterflies using psi_rev_full[((2^OUT + g) << s') + j].
template <int LOGN, class MulMod>
inline void build_tw_tables_sub(uint64_t q, uint64_t psi, uint64_t psi_inv, uint64_t n_inv,
                                MulMod mulmod, int OUT, unsigned g, TW* fwd, TW* inv)
{
  using G = Geo<LOGN>;
  const int FULL = LOGN + OUT;
  const int NF = 1 << FULL;
  // pw[i] = psi^i, ipw[i] = psi^-i
  uint64_t* pw = new uint64_t[NF];
  uint64_t* ipw = new uint64_t[NF];
  pw[0] = ipw[0] = 1;
  for (int i = 1; i < NF; i++) {
    pw[i] = mulmod(pw[i - 1], psi, q);
    ipw[i] = mulmod(ipw[i - 1], psi_inv, q);
  }
  auto mk = [&](uint64_t w) {
    TW t;
    t.w = w;
    t.wp = (uint64_t)((((unsigned __int128)w) << 64) / q);
    return t;
  };
  // local index (2^s' + j) -> exponent of psi
  auto src = [&](unsigned idx) {
    int sp = 31 - __builtin_clz(idx);
    unsigned j = idx - (1u << sp);
    unsigned full = (((1u << OUT) + g) << sp) + j;
    return brev_bits(full, FULL);
  };
  for (int i = 0; i < G::TW_TOTAL; i++) {
    fwd[i] = mk(0);
    inv[i] = mk(0);
  }
  for (int s = 0; s < 5; s++)
    for (int k = 0; k < (1 << s); k++) {
      unsigned e = src(twA_src<LOGN>(s, k));
      fwd[(1 << s) - 1 + k] = mk(pw[e]);
      inv[(1 << s) - 1 + k] = mk(ipw[e]);
    }
  // inverse: slot 0 = S0 * N^-1, slot 31 = N^-1
  inv[0] = mk(mulmod(ipw[src(1)], n_inv, q));
  inv[31] = mk(n_inv);
  for (int sp = 0; sp < 5; sp++)
    for (int k = 0; k < (1 << sp); k++)
      for (unsigned hip = 0; hip < 32; hip++) {
        unsigned e = src(twB_src<LOGN>(sp, k, hip));
        fwd[G::TWB + ((1 << sp) - 1 + k) * 32 + hip] = mk(pw[e]);
        inv[G::TWB + ((1 << sp) - 1 + k) * 32 + hip] = mk(ipw[e]);
      }
  for (int sp = 0; sp < G::LC; sp++)
    for (int k = 0; k < (1 << sp); k++)
      for (unsigned u = 0; u < 1024; u++) {
        unsigned e = src(twC_src<LOGN>(sp, k, u));
        fwd[G::TWC + ((1 << sp) - 1 + k) * 1024 + u] = mk(pw[e]);
        inv[G::TWC + ((1 << sp) - 1 + k) * 1024 + u] = mk(ipw[e]);
      }
  delete[] pw;
  delete[] ipw;
}

template <int LOGN, class MulMod>
inline void build_tw_tables(uint64_t q, uint64_t psi, uint64_t psi_inv, uint64_t n_inv,
                            MulMod mulmod, TW* fwd, TW* inv)
{
  build_tw_tables_sub<LOGN>(q, psi, psi_inv, n_inv, mulmod, 0, 0u, fwd, inv);
}
// The same table in Proth form: entry i = w_i 2^64 mod q (8 bytes), positions unchanged.
inline TWM tw_mont_form(uint64_t w, uint64_t q) { return (uint64_t)((((unsigned __int128)w) << 64) % q); }
inline void tw_tables_to_mont(const TW* t, int n, uint64_t q, TWM* out)
{
  for (int i = 0; i < n; i++)
    out[i] = tw_mont_form(t[i].w, q);
}

// ---------------------------------------------------------------------
// Small transforms (N = 2^logn <= 4096): plain bit-reversed psi tables,
//   fwd[idx] = psi^{brev_logn(idx)}, inv[idx] = psi^{-brev_logn(idx)}, 1 <= idx < N,
//   inv[0] = N^-1.  Used by ntt_small_kernel (one workgroup, row resident in LDS).
// ---------------------------------------------------------------------
template <class MulMod>
inline void build_tw_small(int logn, uint64_t q, uint64_t psi, uint64_t psi_inv, uint64_t n_inv,
                           MulMod mulmod, TW* fwd, TW* inv)
{
  const int N = 1 << logn;
  uint64_t* pw = new uint64_t[N];
  uint64_t* ipw = new uint64_t[N];
  pw[0] = ipw[0] = 1;
  for (int i = 1; i < N; i++) {
    pw[i] = mulmod(pw[i - 1], psi, q);
    ipw[i] = mulmod(ipw[i - 1], psi_inv, q);
  }
  auto mk = [&](uint64_t w) {
    TW t;
    t.w = w;
    t.wp = (uint64_t)((((unsigned __int128)w) << 64) / q);
    return t;
  };
  fwd[0] = mk(0);
  inv[0] = mk(n_inv);
  for (int idx = 1; idx < N; idx++) {
    unsigned e = brev_bits((unsigned)idx, logn);
    fwd[idx] = mk(pw[e]);
    inv[idx] = mk(ipw[e]);
  }
  delete[] pw;
  delete[] ipw;
}

}  // namespace hx

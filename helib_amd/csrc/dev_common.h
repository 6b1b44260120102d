// dev_common.h -- device-visible constants and 64-bit modular arithmetic for
// gfx950.  All values are canonical residues of word-sized primes q < 2^60
// (HElib: q < 2^60, src/macro.h:21); arithmetic is exact integer (no MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntt_core.h"

namespace hx {

constexpr int MAX_ROWS = 160;  // rows addressable by one launch descriptor

// Per-prime constants, one entry per Context::moduli index, resident in HBM.
struct PrimeDev {
  uint64_t q;
  uint64_t q2;     // 2q
  uint64_t mu;     // floor(2^(2k) / q), k = bitlen(q)   (Barrett, 128-bit products)
  uint64_t mu64;   // floor(2^64 / q)                     (Barrett, 64-bit values)
  uint32_t k;      // bitlen(q)
  uint32_t proth;  // 1: q = 1 (mod 2^32) and the row tables below hold 8-byte Proth-form entries (ntt_core.h, ArProth)
  uint64_t mu63;   // floor(2^(63+k) / q)                 (red128_q8: 128-bit sums below 8 q^2)
  // power-of-two NTT tables (ntt_core.h layout) as offsets, in TW units, into the
  // context's single twiddle arena: the arena base is a kernel ARGUMENT so the
  // compiler addresses it as global memory (scalar loads for uniform entries)
  // instead of through a flat pointer fetched from memory.
  uint64_t tw_fwd_off;
  uint64_t tw_inv_off;
  uint64_t r2;     // 2^128 mod q: a Proth-form constant times this (mont_mul) carries one more 2^64 (conv_kernels.hip)
  // sub-transform entries of the N = 2^15 forward half-row kernel only (ntt_kernels.hip ntt_row_half15_kernel): the
  // first stage's twiddle psi^(N/2) as a Shoup pair and times 2^64
  uint64_t half_t1, half_t1p, half_t1m;
};

// launch descriptor passed BY VALUE (kernel-arg segment): row r of the
// operand uses prime p[r % period]; b_row[r] = matching row of the second
// operand (element-wise binary ops).
struct RowMap {
  uint16_t p[MAX_ROWS];
};
// NTT launch descriptor: the i-th transformed row is buffer row `row[i]`, prime `prime[i]`.
struct NttRows {
  uint16_t row[MAX_ROWS];
  uint16_t prime[MAX_ROWS];
};

// Fused DoubleCRT::scaleDownToSet for ONE dropped prime qd (src/DoubleCRT.cpp:1464-1516):
//   prep  (IO of the inverse transform of the dropped row): per coefficient x in [0,qd) ->
//         delta = x - qd*S,  S = [x > (qd-1)/2] + balanced((delta0 mod p) * qd^-1 mod p)
//         (centring :1098-1099, ptxtSpace correction :1485-1508); stores x and S.
//   apply (IO of the forward transform of every kept row r): loads delta * qd^-1 mod q_r =
//         x*inv_r - S  (qd*inv_r = 1), and its store is  c_r <- c_r*inv_r - NTT(.)  mod q_r.
struct ModDownPrep {
  uint64_t* xs;        // [batch][N]
  int64_t* S;          // [batch][N]
  uint64_t half;       // (qd-1)/2
  uint64_t ptxt;       // 0/1: no correction
  uint64_t ptxt_mu64, ptxt_mu, qd_mod_p, qdinv_mod_p;
  uint32_t ptxt_k, has_up;
  TW upS, upN;         // fused mod-up (the dropped row times F = prod(added primes)): the last
                       // inverse stage's twiddles with F folded in, F*S0*N^-1 and F*N^-1
                       // (.wp = the companion word of the dropped prime's arithmetic: Shoup's quotient, or the
                       // Proth form w 2^64 mod qd when PrimeDev::proth -- tw_companion() in engine.hip)
  uint64_t qd;
  uint64_t poly_stride;  // words between the x blocks of consecutive polys (0: batch*N, the
                         // single-prime layout [poly][batch][N]; the several-primes path keeps
                         // [poly][dropped prime][batch][N] and launches once per dropped prime)
};
// several dropped primes in ONE prep launch (scale_down_multi_fused): workgroup block j of
// polys.n * batch workgroups inverse-transforms dropped row row[j] (prime prime[j]) of every poly
// into x block j; up[2j], up[2j+1] = that prime's F*S0*N^-1 and F*N^-1 when the mod-up is folded in
constexpr int MD_MAXDROP = 16;
struct PrepMulti {
  uint16_t row[MD_MAXDROP];
  uint16_t prime[MD_MAXDROP];
  TW up[2 * MD_MAXDROP];
};
struct ModDownRow {
  TW qdm;              // qd mod q_r
  TW inv;              // qd^-1 mod q_r
  TW cf;               // fused mod-up: F * qd^-1 mod q_r (multiplies c_r instead of inv)
                       // (inv.wp, cf.wp: the companion word of the KEPT row's arithmetic -- Shoup's quotient, or the
                       // Proth form w 2^64 mod q_r when PrimeDev::proth)
  uint64_t cf_r2;      // Proth rows: cf 2^128 mod q_r -- multiplies a data x data product that was reduced by
                       // mont_redc128 (and so carries 2^-64) in the tensor form of the apply kernel; else 0
  uint32_t out_row;
  uint32_t mode;       // 0/1: c_r <- c_r*cf - v (cf = inv or F*inv) ; 2: new row, c_r = 0 (cf = 0)
                       // (|S| < q_r is checked by the host before it takes the fused path)
};
struct ModDownApply {
  const uint64_t* xs;
  const int64_t* S;
  const ModDownRow* rows;  // [launch rows]
  // several dropped primes: delta * P^-1 on the kept primes was written by the basis-extension
  // kernel as coefficient rows [poly][kept row][batch][N]; the transform loads those instead of
  // forming x*inv - S (PLAIN instantiation of the apply kernel)
  const uint64_t* delta;
  uint64_t delta_poly_stride;  // words
};
// Ctxt::tensorProduct folded into the mod-switch that follows it (Ctxt::multiplyBy on fresh ciphertexts:
// multLowLvl's tensorProduct, then reLinearize's dropSmallAndSpecialPrimes, src/Ctxt.cpp:1563-1608, 720-760):
// the three product parts are never materialised on the old prime set -- the mod-down kernels form
// part(1) = a0 b0, part(s) = a0 b1 + a1 b0, part(s^2) = a1 b1 from the operand rows where they consume them.
// Streaming rows (read once / written once) are moved with the non-temporal hint so that the data an XCD's L2
// actually re-uses (key-switching matrix rows shared by the batch, twiddle tables, x / S of the mod-down) stays in it;
// -DHX_NO_NT restores the default policy (A/B).
#if defined(__HIPCC__)
typedef unsigned long long hx_v2u64 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ulonglong2 ld_stream2(const uint64_t* p)
{
#ifdef HX_NO_NT
  return *reinterpret_cast<const ulonglong2*>(p);
#else
  const hx_v2u64 v = __builtin_nontemporal_load(reinterpret_cast<const hx_v2u64*>(p));
  return make_ulonglong2(v.x, v.y);
#endif
}
__device__ __forceinline__ uint64_t ld_stream1(const uint64_t* p)
{
#if defined(HX_NO_NT) || defined(HX_RNS_NO_NT)
  return *p;
#else
  return __builtin_nontemporal_load(p);
#endif
}
__device__ __forceinline__ void st_stream1(uint64_t* p, uint64_t v)
{
#if defined(HX_NO_NT) || defined(HX_RNS_NO_NT)
  *p = v;
#else
  __builtin_nontemporal_store(v, p);
#endif
}
__device__ __forceinline__ void st_stream2(uint64_t* p, ulonglong2 v)
{
#ifdef HX_NO_NT
  *reinterpret_cast<ulonglong2*>(p) = v;
#else
  hx_v2u64 w;
  w.x = v.x;
  w.y = v.y;
  __builtin_nontemporal_store(w, reinterpret_cast<hx_v2u64*>(p));
#endif
}
#endif

struct TensorSrc {
  const uint64_t* a0;
  const uint64_t* a1;
  const uint64_t* b0;
  const uint64_t* b1;
};
// several DoubleCRT objects with the same prime set (the parts of one or two ciphertexts) are
// mod-switched by one pair of launches: xs/S hold [poly][batch][N]
constexpr int MD_MAXPOLY = 8;
struct PolyBases {
  uint64_t* d[MD_MAXPOLY];
  int n;
};
// d[i] for a wave-uniform i without indexing the by-value kernel argument dynamically (that makes
// the compiler copy the struct to scratch memory): a chain of scalar selects.
__host__ __device__ inline uint64_t* poly_base(const PolyBases& p, unsigned i)
{
  uint64_t* r = p.d[0];
  r = i == 1 ? p.d[1] : r;
  r = i == 2 ? p.d[2] : r;
  r = i == 3 ? p.d[3] : r;
  r = i == 4 ? p.d[4] : r;
  r = i == 5 ? p.d[5] : r;
  r = i == 6 ? p.d[6] : r;
  r = i == 7 ? p.d[7] : r;
  return r;
}
struct RowMap2 {
  uint16_t p[MAX_ROWS];
  uint16_t brow[MAX_ROWS];
};
struct RowScalars {
  uint64_t c[MAX_ROWS];   // scalar in [0,q)
  uint64_t cp[MAX_ROWS];  // Shoup precon floor(c*2^64/q)
};

typedef unsigned __int128 u128;

// Read-only tables (basis-extension plans, per-row constants): pointers into the CONSTANT address
// space.  A load through a plain pointer held in a by-value kernel-argument struct is a vector
// global_load even when its address is wave-uniform -- the kernel also stores, so the compiler
// may not assume the table is invariant -- and a loop over targets then waits a full memory
// round trip per iteration (round 1: rns_extend_fast_kernel<8> ran 4x below its instruction count).
// Through address space 4 the same loads are s_load (scalar cache, issued ahead by the compiler).
#define HX_RO __attribute__((address_space(4)))
typedef const HX_RO uint64_t* ro_u64;
typedef const HX_RO uint32_t* ro_u32;
typedef const HX_RO double* ro_f64;
typedef const HX_RO TW* ro_tw;
template <class T>
__host__ __device__ inline const HX_RO T* as_ro(const T* p)
{
  return (const HX_RO T*)(uintptr_t)p;
}
// (a struct cannot be copy-constructed out of another address space: word by word)
__device__ __forceinline__ TW ld_tw(ro_tw p, size_t i)
{
  ro_u64 w = (ro_u64)p;
  TW t;
  t.w = w[2 * i];
  t.wp = w[2 * i + 1];
  return t;
}

__device__ __forceinline__ uint64_t add_mod(uint64_t a, uint64_t b, uint64_t q)
{
  uint64_t s = a + b;
  return s >= q ? s - q : s;
}
__device__ __forceinline__ uint64_t sub_mod(uint64_t a, uint64_t b, uint64_t q)
{
  return a >= b ? a - b : a + q - b;
}
__device__ __forceinline__ uint64_t neg_mod(uint64_t a, uint64_t q) { return a ? q - a : 0; }

// a*b mod q for a,b in [0,q): NTL::MulMod(a,b,q,qinv) (src/DoubleCRT.cpp:331).
// Classical Barrett on the 128-bit product; exact canonical result.
__device__ __forceinline__ uint64_t mul_mod(uint64_t a, uint64_t b, uint64_t q, uint64_t mu,
                                            uint32_t k)
{
  u128 x = (u128)a * b;
  uint64_t xs = (uint64_t)(x >> (k - 1));            // < 2^(k+1)
  uint64_t qh = (uint64_t)(((u128)xs * mu) >> (k + 1));
  uint64_t r = (uint64_t)x - qh * q;                  // < 3q
  r = r >= q ? r - q : r;
  return r >= q ? r - q : r;
}
// x mod q for a 128-bit x < 8*q^2 (sum of up to 8 products of residues), q < 2^60:
// Barrett with the same mu; the quotient estimate is at most 10 short, fixed by
// conditional subtractions.  Used to reduce once per accumulated inner product.
__device__ __forceinline__ uint64_t red128_wide(u128 x, uint64_t q, uint64_t mu, uint32_t k)
{
  uint64_t xs = (uint64_t)(x >> (k - 1));            // < 2^(k+4) <= 2^64
  uint64_t qh = (uint64_t)(((u128)xs * mu) >> (k + 1));
  uint64_t r = (uint64_t)x - qh * q;                  // < 11q < 2^64
  uint64_t q8 = q << 3, q4 = q << 2, q2 = q << 1;
  r = r >= q8 ? r - q8 : r;
  r = r >= q4 ? r - q4 : r;
  r = r >= q2 ? r - q2 : r;
  return r >= q ? r - q : r;
}
// x mod q for any 64-bit x.
__device__ __forceinline__ uint64_t red64(uint64_t x, uint64_t q, uint64_t mu64)
{
  uint64_t qh = __umul64hi(x, mu64);
  uint64_t r = x - qh * q;  // < 2q
  return r >= q ? r - q : r;
}
// x*c mod q with precomputed cp = floor(c*2^64/q): NTL::MulModPrecon.
__device__ __forceinline__ uint64_t mul_shoup(uint64_t x, uint64_t c, uint64_t cp, uint64_t q)
{
  uint64_t h = __umul64hi(x, cp);
  uint64_t r = x * c - h * q;  // [0,2q)
  return r >= q ? r - q : r;
}

}  // namespace hx

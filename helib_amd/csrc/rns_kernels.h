// rns_kernels.h -- element-wise DoubleCRT kernels and the exact RNS
// basis-extension kernel (device side of addPrimes / breakIntoDigits /
// scaleDownToSet).  Included by engine.hip only.
#pragma once
#include "dev_common.h"
#include "rns_types.h"

namespace hx {

// =====================================================================
// element-wise kernels: layout [row][batch][N]; grid = (chunks, rows)
// Each thread handles 2 consecutive words (16-byte accesses) per step.
// HBM-bound: 24 B/element (binary), 16 B/element (unary/scalar).
// =====================================================================
enum EwOp { EW_ADD = 0, EW_SUB = 1, EW_MUL = 2 };

// Device-to-device copies of the engine's own buffers go through this kernel instead of
// hipMemcpyAsync: one route (a shader at HBM speed, visible in a kernel trace) whatever copy
// engine the runtime would have picked.  n words, 16 B per lane and step; dst/src 16-byte aligned
// (rows of the slab pool are).
__global__ void __launch_bounds__(256)
copy_words_kernel(uint64_t* __restrict__ dst, const uint64_t* __restrict__ src, size_t n)
{
  const size_t nvec = n / 2;
  ulonglong2* d = reinterpret_cast<ulonglong2*>(dst);
  const ulonglong2* s = reinterpret_cast<const ulonglong2*>(src);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (size_t)gridDim.x * blockDim.x)
    d[i] = s[i];
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
    dst[n - 1] = src[n - 1];
}

template <int OP>
__global__ void __launch_bounds__(256)
ew_binary_kernel(uint64_t* __restrict__ a, const uint64_t* __restrict__ b, RowMap2 map,
                 size_t row_words /* batch*N */, size_t b_row_words, int b_broadcast, size_t n,
                 const PrimeDev* __restrict__ primes)
{
  const int row = blockIdx.y;
  const PrimeDev pd = primes[map.p[row]];
  const uint64_t q = pd.q;
  ulonglong2* pa = reinterpret_cast<ulonglong2*>(a + (size_t)row * row_words);
  const uint64_t* pb_base = b + (size_t)map.brow[row] * b_row_words;
  const size_t nvec = row_words / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t bi = b_broadcast ? ((2 * i) % n) : (2 * i);
    ulonglong2 x = pa[i];
    ulonglong2 y = *reinterpret_cast<const ulonglong2*>(pb_base + bi);
    if (OP == EW_ADD) {
      x.x = add_mod(x.x, y.x, q);
      x.y = add_mod(x.y, y.y, q);
    } else if (OP == EW_SUB) {
      x.x = sub_mod(x.x, y.x, q);
      x.y = sub_mod(x.y, y.y, q);
    } else {
      x.x = mul_mod(x.x, y.x, q, pd.mu, pd.k);
      x.y = mul_mod(x.y, y.y, q, pd.mu, pd.k);
    }
    pa[i] = x;
  }
}

enum EwSOp { EWS_ADD = 0, EWS_SUB = 1, EWS_MUL = 2, EWS_NEG = 3, EWS_SET = 4, EWS_EXP = 5 };

template <int OP>
__global__ void __launch_bounds__(256)
ew_scalar_kernel(uint64_t* __restrict__ a, RowMap map, RowScalars sc, size_t row_words,
                 const PrimeDev* __restrict__ primes)
{
  const int row = blockIdx.y;
  const PrimeDev pd = primes[map.p[row]];
  const uint64_t q = pd.q;
  const uint64_t c = sc.c[row], cp = sc.cp[row];
  ulonglong2* pa = reinterpret_cast<ulonglong2*>(a + (size_t)row * row_words);
  const size_t nvec = row_words / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (size_t)gridDim.x * blockDim.x) {
    if (OP == EWS_SET) {  // DoubleCRT::operator=(ZZ): every entry = num mod q
      pa[i] = make_ulonglong2(c, c);
      continue;
    }
    ulonglong2 x = pa[i];
    if (OP == EWS_EXP) {  // DoubleCRT::Exp: PowerMod(x, e, q), e = cp (the same for every row)
      uint64_t rx = 1 % q, ry = 1 % q, bx = x.x, by = x.y;
      for (uint64_t e = cp; e; e >>= 1) {
        if (e & 1) {
          rx = mul_mod(rx, bx, q, pd.mu, pd.k);
          ry = mul_mod(ry, by, q, pd.mu, pd.k);
        }
        bx = mul_mod(bx, bx, q, pd.mu, pd.k);
        by = mul_mod(by, by, q, pd.mu, pd.k);
      }
      pa[i] = make_ulonglong2(rx, ry);
      continue;
    }
    if (OP == EWS_ADD) {
      x.x = add_mod(x.x, c, q);
      x.y = add_mod(x.y, c, q);
    } else if (OP == EWS_SUB) {
      x.x = sub_mod(x.x, c, q);
      x.y = sub_mod(x.y, c, q);
    } else if (OP == EWS_MUL) {
      x.x = mul_shoup(x.x, c, cp, q);
      x.y = mul_shoup(x.y, c, cp, q);
    } else {
      x.x = neg_mod(x.x, q);
      x.y = neg_mod(x.y, q);
    }
    pa[i] = x;
  }
}

// (o0, o1) = (a0, a1) * c per row, out of place: parts (1) and (s) of reLinearize
__global__ void __launch_bounds__(256)
scale2_kernel(const uint64_t* __restrict__ a0, const uint64_t* __restrict__ a1,
              uint64_t* __restrict__ o0, uint64_t* __restrict__ o1, RowMap map, RowScalars sc,
              size_t row_words, const PrimeDev* __restrict__ primes)
{
  const int row = blockIdx.y;
  const uint64_t q = primes[map.p[row]].q;
  const uint64_t c = sc.c[row], cp = sc.cp[row];
  const size_t off = (size_t)row * row_words;
  const size_t nvec = row_words / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = off + 2 * i;
    ulonglong2 x = *reinterpret_cast<const ulonglong2*>(a0 + e);
    ulonglong2 y = *reinterpret_cast<const ulonglong2*>(a1 + e);
    x.x = mul_shoup(x.x, c, cp, q);
    x.y = mul_shoup(x.y, c, cp, q);
    y.x = mul_shoup(y.x, c, cp, q);
    y.y = mul_shoup(y.y, c, cp, q);
    *reinterpret_cast<ulonglong2*>(o0 + e) = x;
    *reinterpret_cast<ulonglong2*>(o1 + e) = y;
  }
}

// automorph: out[row][b][j] = in[row][b][perm(j)], perm shared by all rows.
// pow2 m: perm(j) = (((2j+1)*k mod m) - 1)/2 computed in registers;
// general m: perm table built once per call by perm_kernel.
__global__ void __launch_bounds__(256)
perm_build_kernel(uint32_t* __restrict__ perm, const uint32_t* __restrict__ zms,
                  const int32_t* __restrict__ zms_index, uint32_t phim, uint64_t m, uint64_t k)
{
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < phim) {
    uint64_t t = ((uint64_t)zms[j] * k) % m;
    perm[j] = (uint32_t)zms_index[t];
  }
}
__global__ void __launch_bounds__(256)
gather_kernel(uint64_t* __restrict__ out, const uint64_t* __restrict__ in,
              const uint32_t* __restrict__ perm, uint32_t n, size_t nseg /* rows*batch */,
              int pow2, uint64_t m, uint64_t k)
{
  const size_t seg = blockIdx.y;
  const uint64_t* src = in + seg * n;
  uint64_t* dst = out + seg * n;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    uint32_t s;
    if (pow2)
      s = (uint32_t)(((((uint64_t)(2 * j + 1)) * k) & (m - 1)) >> 1);
    else
      s = perm[j];
    dst[j] = src[s];
  }
}

// =====================================================================
// Ctxt::tensorProduct for 2x2 parts, fused (src/Ctxt.cpp:1576-1597), with the
// optional scalar of addPrimesAndScale (src/DoubleCRT.cpp:617-636) applied to
// the parts that point at 1 and s (reLinearize, src/Ctxt.cpp:764-767).
// 56 B/element instead of the reference's 5 separate passes (120 B/element).
// =====================================================================
__device__ __forceinline__ uint64_t red128_q8(u128 S, uint64_t q, uint64_t mu63, uint32_t k);  // (defined below)
__global__ void __launch_bounds__(256)
tensor_kernel(const uint64_t* __restrict__ c0, const uint64_t* __restrict__ c1,
              const uint64_t* __restrict__ d0, const uint64_t* __restrict__ d1,
              uint64_t* __restrict__ o0, uint64_t* __restrict__ o1, uint64_t* __restrict__ o2,
              RowMap map, RowScalars sc, int scale, size_t row_words,
              const PrimeDev* __restrict__ primes)
{
  const int row = blockIdx.y;
  const PrimeDev pd = primes[map.p[row]];
  const uint64_t q = pd.q;
  const uint32_t k = pd.k;
  const uint64_t c = sc.c[row], cp = sc.cp[row];
  const size_t off = (size_t)row * row_words;
  const size_t nvec = row_words / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = off + 2 * i;
    ulonglong2 a0 = *reinterpret_cast<const ulonglong2*>(c0 + e);
    ulonglong2 a1 = *reinterpret_cast<const ulonglong2*>(c1 + e);
    ulonglong2 b0 = *reinterpret_cast<const ulonglong2*>(d0 + e);
    ulonglong2 b1 = *reinterpret_cast<const ulonglong2*>(d1 + e);
    ulonglong2 r0, r1, r2;
    // (red128_q8: the approximate-quotient Barrett, 7 word multiplications; the (s) part takes its two 128-bit
    // products in one reduction -- three reductions per coefficient instead of four classical ones)
    const uint64_t m63 = pd.mu63;
    r0.x = red128_q8((u128)a0.x * b0.x, q, m63, k);
    r0.y = red128_q8((u128)a0.y * b0.y, q, m63, k);
    r1.x = red128_q8((u128)a0.x * b1.x + (u128)a1.x * b0.x, q, m63, k);
    r1.y = red128_q8((u128)a0.y * b1.y + (u128)a1.y * b0.y, q, m63, k);
    r2.x = red128_q8((u128)a1.x * b1.x, q, m63, k);
    r2.y = red128_q8((u128)a1.y * b1.y, q, m63, k);
    if (scale) {
      r0.x = mul_shoup(r0.x, c, cp, q);
      r0.y = mul_shoup(r0.y, c, cp, q);
      r1.x = mul_shoup(r1.x, c, cp, q);
      r1.y = mul_shoup(r1.y, c, cp, q);
    }
    *reinterpret_cast<ulonglong2*>(o0 + e) = r0;
    *reinterpret_cast<ulonglong2*>(o1 + e) = r1;
    *reinterpret_cast<ulonglong2*>(o2 + e) = r2;
  }
}

// =====================================================================
// Ctxt::keySwitchDigits, fused over digits (src/Ctxt.cpp:204-229):
//   out0[r] += sum_d dig[d][r] * kb[d][r] ; out1[r] += sum_d dig[d][r] * ka[d][r]
// dig: [ndig][nall][batch][N]; kb/ka: [ndig][nall][N] (shared by the batch).
// =====================================================================
// Optional own-row reconstruction: digit `owner` of a ctxt row is not stored in `dig`; it is
//   own = (...((c - d_0) * P_0^-1 - d_1) * P_1^-1 ...)          (src/DoubleCRT.cpp:552-556)
// in the evaluation domain, c = the s^2 part (own_src), d_e = the earlier digits' extension
// rows that the accumulation reads anyway -- so those 16 rows are neither written, transformed
// nor re-read.
__device__ __forceinline__ uint64_t red128_q8(u128 S, uint64_t q, uint64_t mu63, uint32_t k);  // (below)
__device__ __forceinline__ uint64_t red128_q8_lazy(u128 S, uint64_t q, uint64_t mu63, uint32_t k);
// waves per SIMD the fast RNS kernels are compiled for (A/B knobs)
#ifndef HX_EXT_WAVES
#define HX_EXT_WAVES 7
#endif
#ifndef HX_BRK_WAVES
#define HX_BRK_WAVES 7
#endif
constexpr int KS_MAXD = 8;
// (occupancy A/B, round 2: capped at 64 VGPRs / 8 waves per SIMD the kernel spills 60 bytes per lane
// inside its loop and the fixed-level multiply drops from 66 k to 52 k mult/s; at 72 VGPRs / 7 waves
// it is the same as uncapped -- 78 VGPRs, 6 waves: gpurun_out/variants.log, profiles/r02_variants_keyswitch_waves.txt)
#ifndef HX_KS_WAVES
#define HX_KS_WAVES
#endif
struct KsFix {
  int64_t owner;        // digit owning this row, -1 for special primes
  TW pinv[KS_MAXD];     // P_e^-1 mod q_row for e < owner
  TW pscale;            // product of the special primes mod q_row (addPrimesAndScale factor)
};

// ND: compile-time digit count (2..4: every digit's words and key words are requested before the first is used
// -- the run-time loop of ND = 0 waits for each digit's loads in turn, one or two 16-byte loads in flight per lane)
template <int ND>
__global__ void __launch_bounds__(256) HX_KS_WAVES
keyswitch_kernel(const uint64_t* __restrict__ dig, const uint64_t* __restrict__ kb,
                 const uint64_t* __restrict__ ka, uint64_t* __restrict__ out0,
                 uint64_t* __restrict__ out1, RowMap2 map, int ndig, int nall, int wrows, int batch,
                 uint32_t n, int accumulate_rows /* rows < this accumulate, others overwrite */,
                 const PrimeDev* __restrict__ primes, const uint64_t* __restrict__ own_src,
                 const KsFix* __restrict__ fix, int lazy, const uint64_t* __restrict__ t0s,
                 const uint64_t* __restrict__ t1s, TensorSrc ts)
{
  // ts.a0 != null (hx_mul_relin): the tensor product is folded in -- for the rows < accumulate_rows the parts
  // (1), (s) and the s^2 part's evaluation rows are formed here from the four operand parts' rows:
  // t0 = a0 b0, t1 = a0 b1 + a1 b0, t2 = a1 b1; no tensor_kernel pass, no product rows in memory.
  const int row = blockIdx.y;
  const PrimeDev pd = primes[map.p[row]];
  const uint64_t q = pd.q, mu = pd.mu;
  const uint32_t k = pd.k;
  const size_t row_words = (size_t)batch * n;
  const size_t nvec = row_words / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t e = 2 * i;            // within the row: b*n + j
    const size_t j = e % n;            // coefficient index (pairs never straddle: n even)
    ulonglong2 acc0, acc1;
    ulonglong2 own = make_ulonglong2(0, 0);
    const int owner = fix ? (int)fix[row].owner : -1;
    ulonglong2 xs[ND ? ND : 1], kbs[ND ? ND : 1], kas[ND ? ND : 1];
    if constexpr (ND > 0) {
#pragma unroll
      for (int d = 0; d < ND; d++) {
        if (d != owner)
          xs[d] = ld_stream2(dig + ((size_t)d * nall + row) * row_words + e);
        const size_t kr = (size_t)d * wrows + map.brow[row];
        kbs[d] = *reinterpret_cast<const ulonglong2*>(kb + kr * n + j);
        kas[d] = *reinterpret_cast<const ulonglong2*>(ka + kr * n + j);
      }
    }
    if (row < accumulate_rows && ts.a0) {
      const size_t o = (size_t)row * row_words + e;
      const ulonglong2 a0 = ld_stream2(ts.a0 + o);
      const ulonglong2 a1 = ld_stream2(ts.a1 + o);
      const ulonglong2 b0 = ld_stream2(ts.b0 + o);
      const ulonglong2 b1 = ld_stream2(ts.b1 + o);
      const TW ps = fix[row].pscale;
      const uint64_t m63 = pd.mu63;
      // (mul_shoup takes any 64-bit operand -- its result is below q (1 + x / 2^64) before the one conditional
      // subtraction -- so the products it scales stay in [0,6q): no conditional subtractions in between)
      acc0.x = mul_shoup(red128_q8_lazy((u128)a0.x * b0.x, q, m63, k), ps.w, ps.wp, q);
      acc0.y = mul_shoup(red128_q8_lazy((u128)a0.y * b0.y, q, m63, k), ps.w, ps.wp, q);
      acc1.x = mul_shoup(red128_q8_lazy((u128)a0.x * b1.x + (u128)a1.x * b0.x, q, m63, k), ps.w, ps.wp, q);
      acc1.y = mul_shoup(red128_q8_lazy((u128)a0.y * b1.y + (u128)a1.y * b0.y, q, m63, k), ps.w, ps.wp, q);
      // the s^2 part of this row: with earlier digits to take off (owner > 0) its first use is own + q - x inside a
      // mul_shoup as well; as digit 0's own row it enters the sums directly and is finished here (uniform branch)
      own.x = red128_q8_lazy((u128)a1.x * b1.x, q, m63, k);
      own.y = red128_q8_lazy((u128)a1.y * b1.y, q, m63, k);
      if (owner <= 0) {
        own.x = csub(csub(csub(own.x, q << 2), q << 1), q);
        own.y = csub(csub(csub(own.y, q << 2), q << 1), q);
      }
    } else if (row < accumulate_rows && t0s) {
      // parts (1),(s) enter scaled by the special primes (Ctxt::keySwitchPart's
      // addPrimesAndScale, src/Ctxt.cpp:816-820), read straight from the unscaled parts
      const TW ps = fix[row].pscale;
      acc0 = ld_stream2(t0s + (size_t)row * row_words + e);
      acc0.x = mul_shoup(acc0.x, ps.w, ps.wp, q);
      acc0.y = mul_shoup(acc0.y, ps.w, ps.wp, q);
      if (t1s) {  // no part pointing at s (a 2-part ciphertext after an automorphism): zero
        acc1 = ld_stream2(t1s + (size_t)row * row_words + e);
        acc1.x = mul_shoup(acc1.x, ps.w, ps.wp, q);
        acc1.y = mul_shoup(acc1.y, ps.w, ps.wp, q);
      } else {
        acc1 = make_ulonglong2(0, 0);
      }
    } else if (row < accumulate_rows) {
      acc0 = ld_stream2(out0 + (size_t)row * row_words + e);
      acc1 = ld_stream2(out1 + (size_t)row * row_words + e);
    } else {
      acc0 = make_ulonglong2(0, 0);
      acc1 = make_ulonglong2(0, 0);
    }
    if (owner >= 0 && !ts.a0)
      own = ld_stream2(own_src + (size_t)row * row_words + e);
    // lazy inner product: 128-bit sums of the D products, ONE Barrett reduction per output word
    // (q < 2^60 and D <= 8 => the sums stay below 2^123)
    u128 s0x = 0, s0y = 0, s1x = 0, s1y = 0;
    const int nd = ND ? ND : ndig;
#pragma unroll
    for (int d = 0; d < nd; d++) {
      const size_t dr = (size_t)d * nall + row;
      ulonglong2 x;
      if (d == owner) {
        x = own;
      } else {
        if constexpr (ND > 0)
          x = xs[d];
        else
          x = ld_stream2(dig + dr * row_words + e);
        if (d < owner) {
          const TW pi = fix[row].pinv[d];
          own.x = mul_shoup(own.x + q - x.x, pi.w, pi.wp, q);   // own in [0,6q), x in [0,q)
          own.y = mul_shoup(own.y + q - x.y, pi.w, pi.wp, q);
        }
      }
      const size_t kr = (size_t)d * wrows + map.brow[row];  // row of W (may cover more primes)
      ulonglong2 b, a;
      if constexpr (ND > 0) {
        b = kbs[d];
        a = kas[d];
      } else {
        b = *reinterpret_cast<const ulonglong2*>(kb + kr * n + j);
        a = *reinterpret_cast<const ulonglong2*>(ka + kr * n + j);
      }
      if (lazy) {
        s0x += (u128)x.x * b.x;
        s0y += (u128)x.y * b.y;
        s1x += (u128)x.x * a.x;
        s1y += (u128)x.y * a.y;
      } else {
        acc0.x = add_mod(acc0.x, mul_mod(x.x, b.x, q, mu, k), q);
        acc0.y = add_mod(acc0.y, mul_mod(x.y, b.y, q, mu, k), q);
        acc1.x = add_mod(acc1.x, mul_mod(x.x, a.x, q, mu, k), q);
        acc1.y = add_mod(acc1.y, mul_mod(x.y, a.y, q, mu, k), q);
      }
    }
    if (lazy) {
      // (red128_q8: approximate-quotient Barrett with 7 word multiplications -- the classical form
      // with its runtime 128-bit shifts and four compare-subtract steps was a quarter of this kernel's
      // instructions)
      // (the accumulator rides in the same sum: D q^2 + q stays far inside red128_q8's domain for D <= 7)
      acc0.x = red128_q8(s0x + acc0.x, q, pd.mu63, k);
      acc0.y = red128_q8(s0y + acc0.y, q, pd.mu63, k);
      acc1.x = red128_q8(s1x + acc1.x, q, pd.mu63, k);
      acc1.y = red128_q8(s1y + acc1.y, q, pd.mu63, k);
    }
    st_stream2(out0 + (size_t)row * row_words + e, acc0);
    st_stream2(out1 + (size_t)row * row_words + e, acc1);
  }
}

// =====================================================================
// Exact RNS basis extension (device side of DoubleCRT::addPrimes,
// breakIntoDigits and scaleDownToSet).  Per coefficient:
//   1. Garner mixed-radix digits a_k of v = CRT(x_0..x_{n-1}) in [0,P)
//   2. centred: neg = (v > (P-1)/2)  <=>  toPoly's "tmp >= prod_half"
//      (src/DoubleCRT.cpp:1056-1059,1098-1099), compared digit-wise
//   3. for every target prime t: (sum_k a_k*W[t][k] - neg*P) mod t
// Integer-only and exact -- no floating-point quotient estimate (the
// reference's double estimate at :1085-1096 is corrected to the same value).
// Plan tables (uniform -> scalar loads), all uint64:
// =====================================================================
// (struct ExtPlanDev: rns_types.h)

// one target of the fast kernels, read from ExtPlanDev::tgt_pack (record of 10 + 2N words: header, the N multipliers,
// their Shoup companions, 2^64 mod t with its companion): header and multipliers are loaded
// unconditionally at the top of the iteration and pinned in SGPRs (the empty asm keeps the compiler
// from sinking a load into the branch that uses it, where it would wait for it alone) -- the
// compiler merges them into s_load_dwordx8/x16 with ONE wait per target.  The Shoup companions
// (wp, upd) are only read on the paths that need them.
template <int N>
struct TgtRec {
  static constexpr bool WIDE_HDR = N <= 8;  // whole 8-word header in one s_load_dwordx16 (SGPR budget)
  ro_u64 r;
  uint64_t q_, pmod_, mu63_, mu64_, fl_, updw_, updp_;
  uint64_t w_[N];
  static constexpr int STRIDE = 10 + 2 * N;   // header, N multipliers, their N Shoup companions, 2^64 mod t and its companion
  __device__ __forceinline__ TgtRec(ro_u64 pack, int t) : r(pack + (size_t)t * STRIDE)
  {
    q_ = r[0];
    pmod_ = r[1];
    mu63_ = r[2];
    mu64_ = r[3];
    fl_ = r[4];
    if constexpr (WIDE_HDR) {
      updw_ = r[5];
      updp_ = r[6];
    }
#pragma unroll
    for (int k = 0; k < N; k++)
      w_[k] = r[8 + k];
    asm volatile("" : "+s"(q_), "+s"(pmod_), "+s"(mu63_), "+s"(mu64_), "+s"(fl_));
    if constexpr (WIDE_HDR)
      asm volatile("" : "+s"(updw_), "+s"(updp_));
#pragma unroll
    for (int k = 0; k < N; k++)
      asm volatile("" : "+s"(w_[k]));
  }
  __device__ __forceinline__ uint64_t q() const { return q_; }
  __device__ __forceinline__ uint64_t pmod() const { return pmod_; }
  __device__ __forceinline__ uint64_t mu63() const { return mu63_; }
  __device__ __forceinline__ uint64_t mu64() const { return mu64_; }
  __device__ __forceinline__ uint32_t k() const { return (uint32_t)fl_ & 0xffu; }
  __device__ __forceinline__ bool lazy() const { return ((uint32_t)fl_ >> 8) & 1u; }
  __device__ __forceinline__ bool chunk7() const { return ((uint32_t)fl_ >> 9) & 1u; }
  // Proth-form target (ExtPlanDev::ginv_m): the multipliers w(k) are W 2^64 mod q, and the slots its reductions do not
  // need hold (q - P mod q) 2^64 mod q (for mu63) and P^-1 2^64 mod q (for 2^64 mod q)
  __device__ __forceinline__ bool mont() const { return ((uint32_t)fl_ >> 10) & 1u; }
  __device__ __forceinline__ uint64_t negp_m() const { return mu63_; }
  __device__ __forceinline__ uint64_t upd_m() const { return r[8 + 2 * N]; }
  __device__ __forceinline__ TW upd() const
  {
    TW t;
    if constexpr (WIDE_HDR) {
      t.w = updw_;
      t.wp = updp_;
    } else {
      t.w = r[5];
      t.wp = r[6];
    }
    return t;
  }
  __device__ __forceinline__ uint64_t mu() const { return r[7]; }
  // 2^64 mod q with its Shoup companion, for red128_any (the slots behind the multipliers)
  __device__ __forceinline__ TW r64() const
  {
    TW t;
    t.w = r[8 + 2 * N];
    t.wp = r[8 + 2 * N + 1];
    return t;
  }
  __device__ __forceinline__ uint64_t w(int k) const { return w_[k]; }
  // the Shoup companions of the multipliers, all at once (the non-lazy path)
  __device__ __forceinline__ void load_wp(uint64_t (&wp)[N]) const
  {
#pragma unroll
    for (int k = 0; k < N; k++)
      wp[k] = r[8 + N + k];
#pragma unroll
    for (int k = 0; k < N; k++)
      asm volatile("" : "+s"(wp[k]));
  }
};

// sum_k a_k * W[k]  mod q for a target whose plan says the lazy 128-bit form is exact
template <int NMAX>
__device__ __forceinline__ uint64_t mixed_radix_residue_lazy(const uint64_t (&a)[NMAX], ro_tw Wt, int n,
                                                             uint64_t q, uint64_t mu, uint32_t k)
{
  u128 acc = 0;
#pragma unroll
  for (int i = 0; i < NMAX; i++)
    if (i < n)
      acc += (u128)a[i] * ld_tw(Wt, i).w;
  return red128_wide(acc, q, mu, k);
}

// (EXT_MAXSRC, struct ExtArgs: rns_types.h)

// value / P in [0,1) from the mixed-radix digits (value = a_0 + a_1 q_0 + a_2 q_0 q_1 + ...):
// (((a_0/q_0 + a_1)/q_1 + a_2)/q_2 ...)/q_(n-1), the most significant digit entering last
template <int NMAX>
__device__ __forceinline__ double mixed_radix_fraction(const uint64_t (&a)[NMAX], ro_u64 q, int n)
{
  double acc = 0;
#pragma unroll
  for (int k = 0; k < NMAX; k++)
    if (k < n)
      acc = ((double)a[k] + acc) / (double)q[k];
  return acc;
}

template <int NMAX>
__device__ __forceinline__ void rns_extend_one(const ExtPlanDev& P, const ExtArgs& A, size_t row_words, size_t i);

// A.redo given: the listed coefficients only (grid-stride over the list) -- the Garner pass behind the HPS-form
// rns_extend_wide_kernel, whose untrusted lanes write nothing but their index
template <int NMAX>
__global__ void __launch_bounds__(256)
rns_extend_kernel(ExtPlanDev P, ExtArgs A, size_t row_words /* batch*N */)
{
  if (A.redo) {
    const uint32_t n = A.redo[0];
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
      rns_extend_one<NMAX>(P, A, row_words, A.redo[1 + j]);
    return;
  }
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= row_words)
    return;
  rns_extend_one<NMAX>(P, A, row_words, i);
}
template <int NMAX>
__device__ __forceinline__ void rns_extend_one(const ExtPlanDev& P, const ExtArgs& A, size_t row_words, size_t i)
{
  const int n = P.n;
  uint64_t a[NMAX];
  // ---- load + Garner ----
#pragma unroll
  for (int k = 0; k < NMAX; k++) {
    if (k < n) {
      uint64_t x = A.src[(size_t)A.src_row[k] * row_words + i];
      if (A.own_dst_row[k] != 0xffff)
        A.dst[(size_t)A.own_dst_row[k] * row_words + i] = x;
      const uint64_t pk = P.src_q[k], mk = P.src_mu64[k];
#pragma unroll
      for (int l = 0; l < NMAX; l++) {
        if (l < k) {
          uint64_t al = P.garner_cs ? (a[l] >= pk ? a[l] - pk : a[l]) : red64(a[l], pk, mk);
          uint64_t d = sub_mod(x, al, pk);
          TW g = ld_tw(P.ginv, k * n + l);
          x = mul_shoup(d, g.w, g.wp, pk);
        }
      }
      a[k] = x;
    }
  }
  // ---- centring: v > (P-1)/2 by mixed-radix comparison, top digit first ----
  int cmp = 0;  // sign of (v - half)
#pragma unroll
  for (int k = NMAX - 1; k >= 0; k--) {
    if (k < n && cmp == 0) {
      uint64_t h = P.half[k];
      cmp = a[k] > h ? 1 : (a[k] < h ? -1 : 0);
    }
  }
  const bool neg = cmp > 0;

  // ---- BGV: make delta divisible by ptxtSpace (src/DoubleCRT.cpp:1485-1508) ----
  bool dm_nonzero = false, dm_negative = false;
  uint64_t dm_abs = 0;
  if (P.ptxt > 1) {
    const uint64_t p = P.ptxt;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < NMAX; k++) {
      if (k < n) {
        TW w = ld_tw(P.Wp, k);
        acc += shoup_lazy(a[k], w, p);  // each < 2p
        if ((k & 3) == 3)
          acc = red64(acc, p, P.ptxt_mu64);
      }
    }
    uint64_t r = red64(acc, p, P.ptxt_mu64);
    if (neg)
      r = sub_mod(r, P.pmod_ptxt, p);  // delta mod p, non-negative (NTL rem)
    if (r != 0) {
      uint64_t dm = mul_mod(r, P.pinv_ptxt, p, P.ptxt_mu, P.ptxt_k);
      const uint64_t p_over_2 = p >> 1;
      bool sub_p = dm > p_over_2 || (((p & 1) == 0) && dm == p_over_2 && neg);
      dm_nonzero = true;
      dm_negative = sub_p;
      dm_abs = sub_p ? p - dm : dm;  // |delta_i_modP| after balancing
    }
  }

  if (A.frac) {
    double fr = mixed_radix_fraction<NMAX>(a, P.src_q, n) - (neg ? 1.0 : 0.0);
    if (dm_nonzero)
      fr += dm_negative ? (double)dm_abs : -(double)dm_abs;
    A.frac[i] = fr;
  }

  // ---- residues modulo every target prime ----
  auto residue = [&](int t) -> uint64_t {
    const uint64_t q = P.tgt_q[t], mu64 = P.tgt_mu64[t];
    ro_tw Wt = P.W + (size_t)t * n;
    uint64_t r;
    if (P.tgt_lazy[t]) {
      r = mixed_radix_residue_lazy<NMAX>(a, Wt, n, q, P.tgt_mu[t], P.tgt_k[t]);
    } else {
      uint64_t acc = 0;
#pragma unroll
      for (int k = 0; k < NMAX; k++) {
        if (k < n) {
          acc += shoup_lazy(a[k], ld_tw(Wt, k), q);  // each < 2q < 2^61 (q < 2^60)
          if ((k & 3) == 3)
            acc = red64(acc, q, mu64);
        }
      }
      r = red64(acc, q, mu64);
    }
    if (neg)
      r = sub_mod(r, P.pmod[t], q);
    if (dm_nonzero) {
      // delta -= diffProd * delta_i_modP
      uint64_t corr = dm_abs;
      if (!P.corr_unit)
        corr = mul_mod(P.pmod[t], red64(dm_abs, q, mu64), q, P.tgt_mu[t], P.tgt_k[t]);
      r = dm_negative ? add_mod(r, corr, q) : sub_mod(r, corr, q);
    }
    return r;
  };
  // (preloading the read-modify-write rows up front measured slower than this simple loop)
  for (int t = 0; t < P.nt; t++) {
    const uint64_t q = P.tgt_q[t];
    const uint64_t r = residue(t);
    if (A.dst_row[t] != 0xffff)
      A.dst[(size_t)A.dst_row[t] * row_words + i] = r;
    if (A.upd_row[t] != 0xffff) {
      uint64_t* u = A.upd + (size_t)A.upd_row[t] * row_words + i;
      const TW pinv = ld_tw(P.upd, t);
      *u = mul_shoup(sub_mod(*u, r, q), pinv.w, pinv.wp, q);
    }
  }
}

// =====================================================================
// breakIntoDigits, all digits in ONE pass over the coefficients
// (src/DoubleCRT.cpp:479-561).  Each thread owns one coefficient of one batch
// element: it reads the L residues of the s^2 part once into a private LDS
// column, then for digit d = 0,1,..: Garner on the digit's own residues, centred
// lift, residues modulo every other prime written straight to the digit block,
// and the later digits' residues updated in LDS ("digits[j] -= digits[i];
// digits[j] /= pi").  Algorithmic traffic: 8*N*(L + D*(L+K) - L) bytes.
// Digits must be contiguous runs of the operand's rows (HElib's digits are).
// =====================================================================
constexpr int BRK_THREADS = 128;
struct BreakArgs {
  const uint64_t* src;   // coefficient rows of the operand, [L][batch][N]
  uint64_t* dst;         // digit blocks, [ndig][nall][batch][N] (own rows are not written)
  int L, nall, ndig;
  int off[KS_MAXD + 1];  // digit d = rows [off[d], off[d+1])
  ExtPlanDev plan[KS_MAXD];  // plan d: sources = digit d's primes, targets = all other rows, ascending
  double* frac;          // optional [ndig][batch][N]: centred digit / P_d (for its canonical norm,
                         // src/DoubleCRT.cpp:538-545)
  uint32_t* redo;        // as ExtArgs::redo
};

// rows of the first digit that the fast kernel does not keep in LDS (host and device agree on this)
__host__ __device__ inline int break_fast_n0(const BreakArgs& A)
{
  return (A.ndig > 1 && A.off[0] == 0) ? A.off[1] : 0;
}

template <int NMAX>
__global__ void __launch_bounds__(BRK_THREADS)
break_digits_kernel(BreakArgs A, size_t row_words)
{
  extern __shared__ __attribute__((aligned(16))) uint64_t xs[];  // [L][BRK_THREADS]
  const unsigned tid = threadIdx.x;
  const size_t i = (size_t)blockIdx.x * BRK_THREADS + tid;
  if (i >= row_words)
    return;
  for (int r = 0; r < A.L; r++)
    xs[r * BRK_THREADS + tid] = A.src[(size_t)r * row_words + i];
  for (int d = 0; d < A.ndig; d++) {
    const ExtPlanDev& P = A.plan[d];
    const int n = P.n, off = A.off[d];
    uint64_t a[NMAX];
#pragma unroll
    for (int k = 0; k < NMAX; k++) {
      if (k < n) {
        uint64_t x = xs[(off + k) * BRK_THREADS + tid];
        const uint64_t pk = P.src_q[k], mk = P.src_mu64[k];
#pragma unroll
        for (int l = 0; l < NMAX; l++) {
          if (l < k) {
            uint64_t al = P.garner_cs ? (a[l] >= pk ? a[l] - pk : a[l]) : red64(a[l], pk, mk);
            TW g = ld_tw(P.ginv, k * n + l);
            x = mul_shoup(sub_mod(x, al, pk), g.w, g.wp, pk);
          }
        }
        a[k] = x;
      }
    }
    int cmp = 0;
#pragma unroll
    for (int k = NMAX - 1; k >= 0; k--) {
      if (k < n && cmp == 0) {
        uint64_t h = P.half[k];
        cmp = a[k] > h ? 1 : (a[k] < h ? -1 : 0);
      }
    }
    const bool neg = cmp > 0;
    if (A.frac)
      A.frac[(size_t)d * row_words + i] = mixed_radix_fraction<NMAX>(a, P.src_q, n) - (neg ? 1.0 : 0.0);
    uint64_t* dd = A.dst + (size_t)d * A.nall * row_words + i;
    for (int t = 0; t < P.nt; t++) {
      const int r = t < off ? t : t + n;  // row of target t in the all-rows order
      const uint64_t q = P.tgt_q[t], mu64 = P.tgt_mu64[t];
      ro_tw Wt = P.W + (size_t)t * n;
      uint64_t v;
      if (P.tgt_lazy[t]) {
        v = mixed_radix_residue_lazy<NMAX>(a, Wt, n, q, P.tgt_mu[t], P.tgt_k[t]);
      } else {
        uint64_t acc = 0;
#pragma unroll
        for (int k = 0; k < NMAX; k++) {
          if (k < n) {
            acc += shoup_lazy(a[k], ld_tw(Wt, k), q);
            if ((k & 3) == 3)
              acc = red64(acc, q, mu64);
          }
        }
        v = red64(acc, q, mu64);
      }
      if (neg)
        v = sub_mod(v, P.pmod[t], q);
      dd[(size_t)r * row_words] = v;
      if (r >= off + n && r < A.L) {
        const TW pinv = ld_tw(P.upd, t);
        uint64_t* u = &xs[r * BRK_THREADS + tid];
        *u = mul_shoup(sub_mod(*u, v, q), pinv.w, pinv.wp, q);
      }
    }
  }
}

// =====================================================================
// breakIntoDigits, fast form (same contract as break_digits_kernel; chosen by the host when every
// plan has fast_ok).  Differences, all exact:
//   * the digit size n is a compile-time constant per digit (switch on n), so the Garner chain
//     and the inner products are straight-line code without per-element branches;
//   * Garner steps and the later-digit updates use shoup4 (approximate high product, [0,4q)) on
//     lazy values; each mixed-radix digit is normalised once;
//   * a target whose plan allows the wide sum (tgt_lazy) accumulates sum_k a_k W_k in three
//     64-bit accumulators over 30-bit limbs -- a = a1 2^30 + a0, W = w1 2^30 + w0 (both < 2^60):
//     acc00 += a0 w0, acc01 += a0 w1 + a1 w0, acc11 += a1 w1, i.e. four v_mad_u64_u32 per term
//     and no carry handling (16 products of < 2^60 fit 64 bits; the W limbs are split on the
//     scalar unit) -- followed by ONE Barrett reduction of acc00 + 2^30 acc01 + 2^60 acc11;
//     other targets sum shoup4 products and normalise once with the 32-bit reciprocal;
//   * the centring correction "- P mod t" enters the sum as + (t - P mod t) before the reduction.
// =====================================================================
// S mod q for S < 8 q^2 (q < 2^60, k = bitlen(q)), mu63 = floor(2^(63+k)/q) < 2^64.
// xt = floor(S / 2^(k-1)) < 2^(k+4) fits 64 bits and xt mu63 / 2^64 > S/q - 2 (S/2^(k+63) < 1 and
// 2^(k-1)/q < 1); the approximate high product loses at most 2 more and the floor 1, so the
// quotient estimate is at most 5 short: r = S - qh q (mod 2^64) lies in [0, 6q) and three
// conditional subtractions finish.  7 word multiplications instead of the 11 of the classical form.
__device__ __forceinline__ uint64_t red128_q8(u128 S, uint64_t q, uint64_t mu63, uint32_t k)
{
  const uint64_t xt = (uint64_t)(S >> (k - 1));
  const uint32_t xl = (uint32_t)xt, xh = (uint32_t)(xt >> 32);
  const uint32_t ml = (uint32_t)mu63, mh = (uint32_t)(mu63 >> 32);
  const uint64_t qh = (uint64_t)xh * mh + __umulhi(xh, ml) + __umulhi(xl, mh);
  uint64_t r = (uint64_t)S - qh * q;
  r = csub(r, q << 2);
  r = csub(r, q << 1);
  return csub(r, q);
}

__device__ __forceinline__ uint64_t norm_any(uint64_t x, uint64_t q, uint32_t mu32);
// the same two reductions stopped before their conditional subtractions: a value congruent to S in [0,6q) resp.
// [0,5q) -- for words whose only reader is a forward row transform that takes lazy input (ntt_row_kernel<., false, 8>)
__device__ __forceinline__ uint64_t red128_q8_lazy(u128 S, uint64_t q, uint64_t mu63, uint32_t k)
{
  const uint64_t xt = (uint64_t)(S >> (k - 1));
  const uint32_t xl = (uint32_t)xt, xh = (uint32_t)(xt >> 32);
  const uint32_t ml = (uint32_t)mu63, mh = (uint32_t)(mu63 >> 32);
  const uint64_t qh = (uint64_t)xh * mh + __umulhi(xh, ml) + __umulhi(xl, mh);
  return (uint64_t)S - qh * q;
}
__device__ __forceinline__ uint64_t red128_any_lazy(u128 S, uint64_t q, TW r64, uint32_t mu32)
{
  return shoup4((uint64_t)(S >> 64), r64, 0 - q) + norm_any((uint64_t)S, q, mu32);
}
// S mod q for ANY S < 2^127 and any prime q in (2^32, 2^60): S = H 2^64 + Lo; H (2^64 mod q) as an approximate Shoup
// product in [0,4q) (shoup4 takes any 64-bit H), Lo normalised with the 32-bit reciprocal, the sum in [0,5q) finished
// by three conditional subtractions.  Ten word multiplications -- against seven for red128_q8, whose domain
// (S < 8 q^2) a sum of more than eight same-size terms, or terms from larger primes, leaves.
__device__ __forceinline__ uint64_t red128_any(u128 S, uint64_t q, TW r64, uint32_t mu32)
{
  uint64_t r = shoup4((uint64_t)(S >> 64), r64, 0 - q) + norm_any((uint64_t)S, q, mu32);
  r = csub(r, q << 2);
  r = csub(r, q << 1);
  return csub(r, q);
}

__device__ __forceinline__ uint64_t norm_any(uint64_t x, uint64_t q, uint32_t mu32)
{
  // any 64-bit x -> [0,q), q > 2^32: e = floor(x mu32 / 2^64) is floor(x/q) or one less
  const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
  const uint64_t t = (uint64_t)xh * mu32 + __umulhi(xl, mu32);
  const uint32_t e = (uint32_t)(t >> 32);
  const uint64_t nq = 0 - q;
  uint64_t r = (uint64_t)e * (uint32_t)nq + x;
  r += (uint64_t)(e * (uint32_t)(nq >> 32)) << 32;
  return csub(r, q);
}

// ---- front ends of the fast kernels: the representation the target sums run over ----
// rep[k]: mixed-radix digits (Garner) or the y_k of the HPS form; cnt: how many times P is subtracted
// in the target sums (Garner: 1 if centred negative; HPS: v + that); fr: value / P - [negative]
template <int N>
struct ExtRep {
  uint64_t rep[N];
  uint32_t cnt;
  bool neg;
  double fr;
};
// Garner mixed-radix digits (src/DoubleCRT.cpp:1031-1100 computes the same value by CRT); x may be lazy
// (< 4 p_k).  a_l < p_l < 2 p_k (garner_cs).
template <int N, class Load>
__device__ __forceinline__ void garner_front(const ExtPlanDev& P, Load load, ExtRep<N>& R, bool want_frac)
{
  uint64_t (&a)[N] = R.rep;
  if (P.src_mont) {
    // Proth-form sources: y = v + 2 p_k - a_l in (0, 6 p_k) (v < 4 p_k on load, below 2 p_k after a step), the
    // Montgomery product by p_l^-1 2^64 in (0, p_k (1 + 6/16 + 2^-32)) -- six multiply-adds for shoup4's nine
#pragma unroll
    for (int k = 0; k < N; k++) {
      uint64_t v = load(k);
      const uint64_t pk = P.src_q[k], pk2 = pk + pk;
      const QC qc = make_qc(pk, 0);
#pragma unroll
      for (int l = 0; l < k; l++)
        v = mont_mul(v + pk2 - a[l], P.ginv_m[k * N + l], qc);
      v = csub(v, pk2);
      a[k] = csub(v, pk);
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; k++) {
      uint64_t v = load(k);
      const uint64_t pk = P.src_q[k], npk = 0 - pk, pk2 = pk + pk;
#pragma unroll
      for (int l = 0; l < k; l++)
        v = shoup4(v + pk2 - a[l], ld_tw(P.ginv, k * N + l), npk);
      v = csub(v, pk2);
      a[k] = csub(v, pk);
    }
  }
  // centring: value > (P-1)/2, top digit first
  int cmp = 0;
#pragma unroll
  for (int k = N - 1; k >= 0; k--) {
    const uint64_t h = P.half[k];
    cmp = cmp != 0 ? cmp : (a[k] > h ? 1 : (a[k] < h ? -1 : 0));
  }
  R.neg = cmp > 0;
  R.cnt = R.neg ? 1u : 0u;
  R.fr = 0;
  if (want_frac) {
    double f = 0;
#pragma unroll
    for (int k = 0; k < N; k++)
      f = ((double)a[k] + f) * P.src_rq[k];
    R.fr = f - (R.neg ? 1.0 : 0.0);
  }
}
// HPS form; returns false when this lane's quotient or sign cannot be trusted to the double sum
template <int N, class Load>
__device__ __forceinline__ bool hps_front(const ExtPlanDev& P, Load load, ExtRep<N>& R)
{
  double z = 0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    const uint64_t pk = P.src_q[k];
    uint64_t y = shoup4(load(k), ld_tw(P.hps_inv, k), 0 - pk);  // any 64-bit x -> [0, 4 p_k)
    y = csub(y, pk + pk);
    y = csub(y, pk);
    R.rep[k] = y;
    z += (double)y * P.src_rq[k];
  }
  const double fl = floor(z), f = z - fl, eps = P.hps_eps;
  R.neg = f > 0.5;
  R.cnt = (uint32_t)fl + (R.neg ? 1u : 0u);
  R.fr = f - (R.neg ? 1.0 : 0.0);
  return !(f < eps || f > 1.0 - eps || fabs(f - 0.5) < eps);
}
// The fast kernels exist in two forms (template parameter HPS).  HPS = true: the HPS front end alone -- a lane that
// cannot trust its double-precision quotient appends its coefficient index to the redo list (ExtArgs / BreakArgs
// ::redo: [0] = count, [1..] = indices) and carries on with whatever it has; nothing it writes is read by another
// lane.  HPS = false: Garner; given a redo list it visits the listed coefficients only and overwrites what the first
// form wrote for them (every output is a pure function of the unmodified source rows), given none it is the whole
// kernel as before.  With both front ends in ONE kernel the allocator needed 90-96 VGPRs (round 2,
// profiles/r02_hps_front_end_ab.log); apart they fit the 7-waves-per-SIMD budget the kernels are tuned for.
__device__ __forceinline__ void redo_append(uint32_t* redo, size_t i)
{
  const uint32_t slot = atomicAdd(&redo[0], 1u);
  redo[1 + slot] = (uint32_t)i;
}

// xs holds rows [n0, L) only: the rows of the FIRST digit (n0 of them when it starts at row 0) are
// consumed once, by their own pass, and are read straight from global memory (src0) -- 10 instead of
// 16 LDS rows per thread at L = 16, digits 6/5/5, which is what bounds the resident waves.
// LAZY: the extension words are left in [0,6q) (congruent, not reduced): three conditional subtractions less per
// word; hx_mul_relin / hx_relinearize, whose forward transform of these rows takes lazy input
// (Round 6 tried the column elsewhere to win back the waves its 16 rows cost at the CKKS chain -- five per SIMD instead
// of seven, worth 24 % on the BGV launch: rebuilt from the thread's own stored extension words, 1.6 - 2.3 x slower;
// kept in global memory, + 21 % at the BGV shape and - 2 % at CKKS.  Neither stayed: profiles/r06_ab_digit_kernel_occupancy.json.)
template <int N, bool HPS, bool LAZY>
__device__ __forceinline__ bool break_digit_pass(const ExtPlanDev& P, uint64_t* xs, unsigned tid, int off, int L,
                                                 uint64_t* dd, size_t row_words, double* frac_out, int n0,
                                                 const uint64_t* src0)
{
  const bool from_global = off < n0;   // (uniform; a digit is either entirely below n0 or entirely above)
  auto load = [&](int k) -> uint64_t {   // [0,4 p_k): later digits' rows are updated lazily
    return from_global ? ld_stream1(src0 + (size_t)(off + k) * row_words) : xs[(off + k - n0) * BRK_THREADS + tid];
  };
  ExtRep<N> R;
  bool trusted = true;
  if constexpr (HPS)
    trusted = hps_front<N>(P, load, R);
  else
    garner_front<N>(P, load, R, frac_out != nullptr);
  const uint64_t (&a)[N] = R.rep;
  const uint32_t cnt = R.cnt;
  ro_u64 pack = HPS ? P.tgt_pack_hps : P.tgt_pack;
  if (frac_out)
    *frac_out = R.fr;
  uint32_t a0[N], a1[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    a0[k] = (uint32_t)a[k] & 0x3fffffffu;
    a1[k] = (uint32_t)(a[k] >> 30);
  }
  // ---- residues modulo every other prime ----
  for (int t = 0; t < P.nt; t++) {
    const int r = t < off ? t : t + N;  // row of target t in the all-rows order
    const TgtRec<N> T(pack, t);
    const uint64_t q = T.q();
    const bool mont = T.mont();
    const uint64_t negP = mont ? T.negp_m() : q - T.pmod();   // -P mod t: enters cnt times (centring, and the HPS quotient)
    const QC qc = make_qc(q, 0);   // (scalar; the Proth-form branches read qh and c1 only)
    uint64_t v;
    if (mont || T.lazy()) {
      // (every limb product is below 2^60 whatever the target's size: the three-accumulator form holds the 2N <= 16
      // middle products of any target)
      uint64_t c00 = 0, c01 = 0, c11 = 0;
#pragma unroll
      for (int k = 0; k < N; k++) {
        const uint64_t w = T.w(k);
        const uint32_t w0 = (uint32_t)w & 0x3fffffffu, w1 = (uint32_t)(w >> 30);
        c00 += (uint64_t)a0[k] * w0;
        c01 += (uint64_t)a0[k] * w1;
        c01 += (uint64_t)a1[k] * w0;
        c11 += (uint64_t)a1[k] * w1;
      }
      const u128 S = (u128)cnt * negP + c00 + ((u128)c01 << 30) + ((u128)c11 << 60);
      if (mont) {
        // S < (sum_k p_k + cnt) q < 2^124: S 2^-64 = sum_k a_k W_k - cnt P (the 2^64 is in the record's words), in
        // (0, q (N/16 + 1 + 2^-32)) -- below 2q for N <= 8, inside the [0,6q) the LAZY readers are declared with
        v = mont_redc128((uint64_t)S, (uint64_t)(S >> 64), qc);
        if (!LAZY) {
          v = csub(v, qc.q2);
          v = csub(v, q);
        }
      } else {
        v = LAZY ? red128_q8_lazy(S, q, T.mu63(), T.k()) : red128_q8(S, q, T.mu63(), T.k());
      }
    } else {
      // a target the terms outgrow (a 56-bit special prime under 60-bit digit primes): the same limb sums,
      // reduced by red128_any -- round 3 summed a Shoup product per term here (nine multiplications each)
      uint64_t c00 = 0, c01 = 0, c10 = 0, c11 = 0;
#pragma unroll
      for (int k = 0; k < N; k++) {
        const uint64_t w = T.w(k);
        const uint32_t w0 = (uint32_t)w & 0x3fffffffu, w1 = (uint32_t)(w >> 30);
        c00 += (uint64_t)a0[k] * w0;
        c01 += (uint64_t)a0[k] * w1;
        c10 += (uint64_t)a1[k] * w0;
        c11 += (uint64_t)a1[k] * w1;
      }
      const u128 S = (u128)cnt * negP + c00 + (((u128)c01 + c10) << 30) + ((u128)c11 << 60);
      v = LAZY ? red128_any_lazy(S, q, T.r64(), (uint32_t)T.mu64()) : red128_any(S, q, T.r64(), (uint32_t)T.mu64());
    }
    st_stream1(dd + (size_t)r * row_words, v);
    if (r >= off + N && r < L) {
      // digits[j] -= digits[i]; digits[j] /= P_i on a later digit's own row (kept lazy, < 4q; v < 6q when LAZY:
      // the offset that keeps the difference positive is 8q then, 12q < 2^64 in all)
      uint64_t* u = &xs[(r - n0) * BRK_THREADS + tid];
      if (mont)   // u < 4q, v < 2q: u + 2q - v in (0, 6q), the product by P^-1 2^64 below q (1 + 6/16 + 2^-32) < 2q
        *u = mont_mul(*u + qc.q2 - v, T.upd_m(), qc);
      else
        *u = shoup4(*u + (LAZY ? q << 3 : q) - v, T.upd(), 0 - q);
    }
  }
  return trusted;
}

// =====================================================================
// rns_extend_kernel in the fast form (same contract, chosen by the host when the plan has
// fast16_ok): N compile-time, Garner with shoup4 on lazy values, residues by 30-bit-limb
// accumulation -- the whole sum at once (tgt_lazy), or seven terms at a time with the previous
// remainder carried (tgt_chunk7) when sum_k p_k is just above 8 q_t, as when as many same-size
// primes are dropped as a 60-bit accumulator scheme was sized for plus one -- otherwise shoup4
// products normalised once.  The plaintext-space correction, the fraction and the stores are the
// generic kernel's, so every output word and every fraction is the same.
// =====================================================================
template <int N, bool HPS>
__device__ __forceinline__ void rns_extend_fast_one(const ExtPlanDev& P, const ExtArgs& A, size_t row_words, size_t i);

template <int N, bool HPS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(N <= 11 ? HX_EXT_WAVES : 4)))
rns_extend_fast_kernel(ExtPlanDev P, ExtArgs A, size_t row_words)
{
  if constexpr (!HPS) {
    if (A.redo) {   // the listed coefficients only (grid-stride over the list)
      const uint32_t n = A.redo[0];
      for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
        rns_extend_fast_one<N, false>(P, A, row_words, A.redo[1 + j]);
      return;
    }
  }
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= row_words)
    return;
  rns_extend_fast_one<N, HPS>(P, A, row_words, i);
}
template <int N, bool HPS>
__device__ __forceinline__ void rns_extend_fast_one(const ExtPlanDev& P, const ExtArgs& A, size_t row_words, size_t i)
{
#pragma unroll
  for (int k = 0; k < N; k++)
    if (A.own_dst_row[k] != 0xffff)
      st_stream1(A.dst + (size_t)A.own_dst_row[k] * row_words + i, ld_stream1(A.src + (size_t)A.src_row[k] * row_words + i));
  auto load = [&](int k) -> uint64_t { return ld_stream1(A.src + (size_t)A.src_row[k] * row_words + i); };
  ExtRep<N> R;
  if constexpr (HPS) {
    if (!hps_front<N>(P, load, R))
      redo_append(A.redo, i);
  } else {
    garner_front<N>(P, load, R, A.frac != nullptr);
  }
  const uint64_t (&a)[N] = R.rep;
  const bool neg = R.neg;
  const uint32_t cnt = R.cnt;
  ro_u64 pack = HPS ? P.tgt_pack_hps : P.tgt_pack;
  ro_tw Wp = HPS ? P.Wp_hps : P.Wp;

  // ---- BGV: make delta divisible by ptxtSpace (src/DoubleCRT.cpp:1485-1508) ----
  bool dm_nonzero = false, dm_negative = false;
  uint64_t dm_abs = 0;
  if (P.ptxt > 1) {
    const uint64_t p = P.ptxt;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
      acc += shoup_lazy(a[k], ld_tw(Wp, k), p);  // each < 2p
      if ((k & 3) == 3)
        acc = red64(acc, p, P.ptxt_mu64);
    }
    uint64_t r = red64(acc, p, P.ptxt_mu64);
    if (cnt == 1)
      r = sub_mod(r, P.pmod_ptxt, p);
    else if (cnt > 1)   // (HPS: cnt * (P mod ptxt) < 2^63, hps_ok asks for ptxt < 2^58)
      r = sub_mod(r, red64((uint64_t)cnt * P.pmod_ptxt, p, P.ptxt_mu64), p);
    if (r != 0) {
      uint64_t dm = mul_mod(r, P.pinv_ptxt, p, P.ptxt_mu, P.ptxt_k);
      const uint64_t p_over_2 = p >> 1;
      bool sub_p = dm > p_over_2 || (((p & 1) == 0) && dm == p_over_2 && neg);
      dm_nonzero = true;
      dm_negative = sub_p;
      dm_abs = sub_p ? p - dm : dm;
    }
  }
  if (A.frac) {
    double fr = R.fr;
    if (dm_nonzero)
      fr += dm_negative ? (double)dm_abs : -(double)dm_abs;
    A.frac[i] = fr;
  }
  uint32_t a0[N], a1[N];
#pragma unroll
  for (int k = 0; k < N; k++) {
    a0[k] = (uint32_t)a[k] & 0x3fffffffu;
    a1[k] = (uint32_t)(a[k] >> 30);
  }
  for (int t = 0; t < P.nt; t++) {
    const TgtRec<N> T(pack, t);
    const uint64_t q = T.q();
    const bool mont = T.mont();
    const uint64_t negP = mont ? T.negp_m() : q - T.pmod();   // -P mod t: enters cnt times (centring, and the HPS quotient)
    uint64_t r;
    if (mont || T.lazy()) {
      // N <= 16 products of < 2^60 per accumulator
      uint64_t c00 = 0, c01 = 0, c10 = 0, c11 = 0;
#pragma unroll
      for (int k = 0; k < N; k++) {
        const uint64_t w = T.w(k);
        const uint32_t w0 = (uint32_t)w & 0x3fffffffu, w1 = (uint32_t)(w >> 30);
        c00 += (uint64_t)a0[k] * w0;
        c01 += (uint64_t)a0[k] * w1;
        c10 += (uint64_t)a1[k] * w0;
        c11 += (uint64_t)a1[k] * w1;
      }
      const u128 S = (u128)cnt * negP + c00 + (((u128)c01 + c10) << 30) + ((u128)c11 << 60);
      if (mont)   // Proth-form target (TgtRec::mont): S < (16 2^60 + cnt) q, r in (0, q (N/16 + 1 + 2^-32)) -- below 3q
        r = mont_redc128((uint64_t)S, (uint64_t)(S >> 64), make_qc(q, 0));
      else
        r = red128_q8_lazy(S, q, T.mu63(), T.k());
    } else {
      // the terms outgrow red128_q8's domain (more than eight same-size sources, or sources larger than the
      // target): the same limb sums, reduced by red128_any.  (Round 2 carried a remainder through chunks of seven
      // terms, or summed a Shoup product per term: 150-190 VALU instructions per target where this takes ~100.)
      uint64_t c00 = 0, c01 = 0, c10 = 0, c11 = 0;
#pragma unroll
      for (int k = 0; k < N; k++) {
        const uint64_t w = T.w(k);
        const uint32_t w0 = (uint32_t)w & 0x3fffffffu, w1 = (uint32_t)(w >> 30);
        c00 += (uint64_t)a0[k] * w0;
        c01 += (uint64_t)a0[k] * w1;
        c10 += (uint64_t)a1[k] * w0;
        c11 += (uint64_t)a1[k] * w1;
      }
      const u128 S = (u128)cnt * negP + c00 + (((u128)c01 + c10) << 30) + ((u128)c11 << 60);
      r = red128_any_lazy(S, q, T.r64(), (uint32_t)T.mu64());
    }
    // r in [0,6q).  A.lazy_out (uniform): the reader takes any bound up to 8, so the three conditional subtractions
    // -- 12 of the ~100 instructions of a target -- are left out and the plaintext-space correction adds without
    // reducing ([0,7q))
    const bool lazy_out = A.lazy_out != 0 && A.nu == 0;   // (an in-place update below needs the canonical word)
    if (!lazy_out) {
      r = csub(r, q << 2);
      r = csub(r, q << 1);
      r = csub(r, q);
    }
    if (dm_nonzero) {
      // delta -= diffProd * delta_i_modP
      uint64_t corr = dm_abs;
      if (!P.corr_unit)
        corr = mul_mod(T.pmod(), red64(dm_abs, q, T.mu64()), q, T.mu(), T.k());
      if (lazy_out)
        r += dm_negative ? corr : q - corr;
      else
        r = dm_negative ? add_mod(r, corr, q) : sub_mod(r, corr, q);
    }
    if (A.dst_row[t] != 0xffff)
      st_stream1(A.dst + (size_t)A.dst_row[t] * row_words + i, r);
    if (A.upd_row[t] != 0xffff) {
      uint64_t* u = A.upd + (size_t)A.upd_row[t] * row_words + i;
      const TW pinv = T.upd();
      *u = mul_shoup(sub_mod(*u, r, q), pinv.w, pinv.wp, q);
    }
  }
}

// =====================================================================
// rns_extend_kernel for MANY source primes (16 < n <= NMAX <= 40: the 36-prime digits and the 36 dropped special
// primes of the reference's own benchmark chain, bits = 6400 -- benchmarks/bgv_basic.cpp:247), same contract.
// Garner there is n(n-1)/2 = 630 dependent modular products per coefficient and a non-lazy target costs nine
// word multiplications per term; this form is O(n) in front and four per term:
//   * HPS front end only: y_k = x_k (P/p_k)^-1 mod p_k, quotient and sign from the double-precision sum of
//     y_k / p_k.  A lane that cannot trust them (fraction within hps_eps of 0, 1/2 or 1) appends its coefficient
//     to the redo list and writes NOTHING -- rns_extend_kernel<NMAX> (Garner) then does the listed coefficients
//     from the untouched source rows, in-place updates included.
//   * residues modulo a target: sum_k y_k W_k over 30-bit limbs -- the y limbs stay in registers for all targets,
//     the W limbs come pre-split from the host (one 64-bit scalar word per term: w0 | w1 << 32) -- four
//     v_mad_u64_u32 per term into four 64-bit accumulators, flushed into a 128-bit sum every 16 terms (16 products
//     of < 2^60 fit), then ONE reduction of the whole sum, which is below 2^127 whatever the primes' sizes:
//     S = H 2^64 + Lo  ->  shoup4(H, 2^64 mod t) + norm(Lo) in [0,5t), three conditional subtractions.
// n is a run-time value (uniform guards inside a loop unrolled to NMAX), so one instantiation serves every digit
// size of a chain.
// =====================================================================
// (static_for of ntt_core.h: an unrolled loop by construction -- a `#pragma unroll` over 40 iterations of a large
// body is only partly honoured, and the limb arrays then live in LDS / scratch)
template <int K, int N>
__device__ __forceinline__ void pin_limbs(uint32_t (&a)[N], uint32_t (&b)[N])
{
  if constexpr (K < N) {
    asm volatile("" : "+v"(a[K]), "+v"(b[K]));
    pin_limbs<K + 1, N>(a, b);
  }
}

constexpr int WIDE_THREADS = 1024;   // one workgroup per CU at <= 128 VGPRs: ONE copy of the plan's multipliers in its LDS
// (wide_stride: rns_types.h)

template <int NMAX>   // NMAX a multiple of 4
__global__ void __launch_bounds__(WIDE_THREADS)
rns_extend_wide_kernel(ExtPlanDev P, ExtArgs A, size_t row_words)
{
  // The multipliers of ALL targets (nt x n words of 2 x 30-bit limbs; 107 x 36 x 8 B = 30 KB for a digit of the
  // bits = 6400 chain) are staged in LDS once per workgroup of 1024 coefficients and read from there as broadcast
  // ds_read_b128 -- two terms per read, in-order returns the compiler can wait on one by one.  (First form of this
  // kernel: one s_load_dwordx8 per four terms with a full wait behind each -- nine exposed scalar round trips per
  // target at four waves per SIMD: 341 us per digit where the multiply-adds alone need ~150.)
  extern __shared__ __attribute__((aligned(16))) uint64_t wide_lds[];
  const int n = P.n;
  const int stride = wide_stride(n);
  {
    const int words = P.nt * stride;
    for (int j = (int)threadIdx.x; j < words; j += WIDE_THREADS)
      wide_lds[j] = P.wide_pack[j];
  }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * WIDE_THREADS + threadIdx.x;
  if (i >= row_words)
    return;
  uint32_t a0[NMAX], a1[NMAX];
  double z = 0;
  const uint64_t p = P.ptxt;
  uint64_t pacc = 0;
  static_for<0, NMAX>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    uint32_t lo = 0, hi = 0;
    if (k < n) {
      const uint64_t x = ld_stream1(A.src + (size_t)A.src_row[k] * row_words + i);
      if (A.own_dst_row[k] != 0xffff)
        st_stream1(A.dst + (size_t)A.own_dst_row[k] * row_words + i, x);
      const uint64_t pk = P.src_q[k];
      uint64_t y = shoup4(x, ld_tw(P.hps_inv, k), 0 - pk);  // any 64-bit x -> [0, 4 p_k)
      y = csub(y, pk + pk);
      y = csub(y, pk);
      lo = (uint32_t)y & 0x3fffffffu;
      hi = (uint32_t)(y >> 30);
      z += (double)y * P.src_rq[k];
      if (p > 1) {
        pacc += shoup_lazy(y, ld_tw(P.Wp_hps, k), p);  // each < 2p
        if ((k & 3) == 3)
          pacc = red64(pacc, p, P.ptxt_mu64);
      }
    }
    // (pinned as 32-bit values: the compiler otherwise keeps each limb zero-extended in a register PAIR for the
    // 64-bit multiply-adds below -- 160 VGPRs of limbs at n = 40)
    asm volatile("" : "+v"(lo), "+v"(hi));
    a0[k] = lo;
    a1[k] = hi;
  });
  const double fl = floor(z), f = z - fl, eps = P.hps_eps;
  if (f < eps || f > 1.0 - eps || fabs(f - 0.5) < eps) {
    redo_append(A.redo, i);
    return;
  }
  const bool neg = f > 0.5;
  const uint32_t cnt = (uint32_t)fl + (neg ? 1u : 0u);

  // ---- BGV: make delta divisible by ptxtSpace (src/DoubleCRT.cpp:1485-1508) ----
  bool dm_nonzero = false, dm_negative = false;
  uint64_t dm_abs = 0;
  if (p > 1) {
    uint64_t r = red64(pacc, p, P.ptxt_mu64);
    r = sub_mod(r, red64((uint64_t)cnt * P.pmod_ptxt, p, P.ptxt_mu64), p);   // (cnt (P mod ptxt) < 2^63: ptxt < 2^56)
    if (r != 0) {
      uint64_t dm = mul_mod(r, P.pinv_ptxt, p, P.ptxt_mu, P.ptxt_k);
      const uint64_t p_over_2 = p >> 1;
      bool sub_p = dm > p_over_2 || (((p & 1) == 0) && dm == p_over_2 && neg);
      dm_nonzero = true;
      dm_negative = sub_p;
      dm_abs = sub_p ? p - dm : dm;
    }
  }
  if (A.frac) {
    double fr = f - (neg ? 1.0 : 0.0);
    if (dm_nonzero)
      fr += dm_negative ? (double)dm_abs : -(double)dm_abs;
    A.frac[i] = fr;
  }
  for (int t = 0; t < P.nt; t++) {
    // header from the global copy through the scalar unit (uniform), multipliers from LDS
    ro_u64 rec = P.wide_pack + (size_t)t * (size_t)stride;
    const uint64_t q = rec[0], pmod = rec[1];
    TW R64;
    R64.w = rec[2];
    R64.wp = rec[3];
    const uint32_t mu32 = (uint32_t)rec[4];
    const ulonglong2* wl = reinterpret_cast<const ulonglong2*>(wide_lds + (size_t)t * (size_t)stride + 8);
    u128 S = (u128)cnt * (q - pmod);   // -P mod t, cnt times: the HPS quotient and the centring
    // (re-pinned every iteration: a zero-extension hoisted out of this loop would double the limbs' registers)
    pin_limbs<0, NMAX>(a0, a1);
    // straight-line over all NMAX terms (NMAX - n <= 3 of them idle: their limbs are zero and the record is padded
    // with zeros) -- no branch between the LDS reads and the multiply-adds, so the reads run ahead of their use
    static_for<0, (NMAX + 15) / 16>([&](auto cc) {
      constexpr int k0 = decltype(cc)::value * 16;
      uint64_t c00 = 0, c01 = 0, c10 = 0, c11 = 0;
      static_for<0, (k0 + 16 < NMAX ? 16 : NMAX - k0) / 2>([&](auto gc) {
        constexpr int k2 = k0 + decltype(gc)::value * 2;
        const ulonglong2 w = wl[k2 / 2];
        const uint64_t w2[2] = {w.x, w.y};
        static_for<0, 2>([&](auto jc) {
          constexpr int j = decltype(jc)::value, k = k2 + j;
          const uint32_t w0 = (uint32_t)w2[j], w1 = (uint32_t)(w2[j] >> 32);
          c00 += (uint64_t)a0[k] * w0;
          c01 += (uint64_t)a0[k] * w1;
          c10 += (uint64_t)a1[k] * w0;
          c11 += (uint64_t)a1[k] * w1;
        });
      });
      S += (u128)c00 + (((u128)c01 + c10) << 30) + ((u128)c11 << 60);
    });
    uint64_t r = red128_any(S, q, R64, mu32);
    if (dm_nonzero) {
      // delta -= diffProd * delta_i_modP
      uint64_t corr = dm_abs;
      if (!P.corr_unit)
        corr = mul_shoup(red64(dm_abs, q, rec[4]), pmod, rec[7], q);
      r = dm_negative ? add_mod(r, corr, q) : sub_mod(r, corr, q);
    }
    if (A.dst_row[t] != 0xffff)
      st_stream1(A.dst + (size_t)A.dst_row[t] * row_words + i, r);
    if (A.upd_row[t] != 0xffff) {
      uint64_t* u = A.upd + (size_t)A.upd_row[t] * row_words + i;
      *u = mul_shoup(sub_mod(*u, r, q), rec[5], rec[6], q);
    }
  }
}

template <bool HPS, bool LAZY>
__device__ __forceinline__ void break_digits_fast_one(const BreakArgs& A, size_t row_words, uint64_t* xs, unsigned tid, size_t i)
{
  const int n0 = break_fast_n0(A);
  for (int r = n0; r < A.L; r++)
    xs[(r - n0) * BRK_THREADS + tid] = ld_stream1(A.src + (size_t)r * row_words + i);
  const uint64_t* src0 = A.src + i;
  bool trusted = true;
  for (int d = 0; d < A.ndig; d++) {
    const ExtPlanDev& P = A.plan[d];
    const int off = A.off[d];
    uint64_t* dd = A.dst + (size_t)d * A.nall * row_words + i;
    double* fo = A.frac ? A.frac + (size_t)d * row_words + i : nullptr;
    bool ok;
    switch (P.n) {
      case 1: ok = break_digit_pass<1, HPS, LAZY>(P, xs, tid, off, A.L, dd, row_words, fo, n0, src0); break;
      case 2: ok = break_digit_pass<2, HPS, LAZY>(P, xs, tid, off, A.L, dd, row_words, fo, n0, src0); break;
      case 3: ok = break_digit_pass<3, HPS, LAZY>(P, xs, tid, off, A.L, dd, row_words, fo, n0, src0); break;
      case 4: ok = break_digit_pass<4, HPS, LAZY>(P, xs, tid, off, A.L, dd, row_words, fo, n0, src0); break;
      case 5: ok = break_digit_pass<5, HPS, LAZY>(P, xs, tid, off, A.L, dd, row_words, fo, n0, src0); break;
      case 6: ok = break_digit_pass<6, HPS, LAZY>(P, xs, tid, off, A.L, dd, row_words, fo, n0, src0); break;
      case 7: ok = break_digit_pass<7, HPS, LAZY>(P, xs, tid, off, A.L, dd, row_words, fo, n0, src0); break;
      default: ok = break_digit_pass<8, HPS, LAZY>(P, xs, tid, off, A.L, dd, row_words, fo, n0, src0); break;
    }
    trusted = trusted && ok;
  }
  if (HPS && !trusted)   // (one entry per coefficient: a wrong digit also spoils the later digits' rows of this lane)
    redo_append(A.redo, i);
}
template <bool HPS, bool LAZY = false>
__global__ void __launch_bounds__(BRK_THREADS) __attribute__((amdgpu_waves_per_eu(HX_BRK_WAVES)))
break_digits_fast_kernel(BreakArgs A, size_t row_words)
{
  extern __shared__ __attribute__((aligned(16))) uint64_t xs[];  // [L - n0][BRK_THREADS]: one private column per thread
  const unsigned tid = threadIdx.x;
  if constexpr (!HPS) {
    if (A.redo) {   // the listed coefficients only
      const uint32_t n = A.redo[0];
      for (size_t j = (size_t)blockIdx.x * BRK_THREADS + tid; j < n; j += (size_t)gridDim.x * BRK_THREADS)
        break_digits_fast_one<false, LAZY>(A, row_words, xs, tid, A.redo[1 + j]);
      return;
    }
  }
  const size_t i = (size_t)blockIdx.x * BRK_THREADS + tid;
  if (i >= row_words)
    return;
  break_digits_fast_one<HPS, LAZY>(A, row_words, xs, tid, i);
}

// (x - y) * c per row: the tail of scaleDownToSet (*this -= delta; *this /= diffProd)
__global__ void __launch_bounds__(256)
sub_scale_kernel(uint64_t* __restrict__ a, const uint64_t* __restrict__ b, RowMap2 map,
                 RowScalars sc, size_t row_words, const PrimeDev* __restrict__ primes)
{
  const int row = blockIdx.y;
  const uint64_t q = primes[map.p[row]].q;
  const uint64_t c = sc.c[row], cp = sc.cp[row];
  ulonglong2* pa = reinterpret_cast<ulonglong2*>(a + (size_t)row * row_words);
  const ulonglong2* pb = reinterpret_cast<const ulonglong2*>(b + (size_t)map.brow[row] * row_words);
  const size_t nvec = row_words / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (size_t)gridDim.x * blockDim.x) {
    ulonglong2 x = pa[i], y = pb[i];
    x.x = mul_shoup(sub_mod(x.x, y.x, q), c, cp, q);
    x.y = mul_shoup(sub_mod(x.y, y.y, q), c, cp, q);
    pa[i] = x;
  }
}

// the same with the minuend elsewhere: d[row] = (x[map.brow[row]] - d[row]) * c, where d holds
// delta on entry -- the kept rows are read from the old slab where they lie and the result becomes
// the polynomial's new, compact storage
__global__ void __launch_bounds__(256)
sub_scale_from_kernel(uint64_t* __restrict__ d, const uint64_t* __restrict__ x, RowMap2 map,
                      RowScalars sc, size_t row_words, const PrimeDev* __restrict__ primes)
{
  const int row = blockIdx.y;
  const uint64_t q = primes[map.p[row]].q;
  const uint64_t c = sc.c[row], cp = sc.cp[row];
  ulonglong2* pd = reinterpret_cast<ulonglong2*>(d + (size_t)row * row_words);
  const ulonglong2* px = reinterpret_cast<const ulonglong2*>(x + (size_t)map.brow[row] * row_words);
  const size_t nvec = row_words / 2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (size_t)gridDim.x * blockDim.x) {
    ulonglong2 y = pd[i], v = px[i];
    v.x = mul_shoup(sub_mod(v.x, y.x, q), c, cp, q);
    v.y = mul_shoup(sub_mod(v.y, y.y, q), c, cp, q);
    pd[i] = v;
  }
}

}  // namespace hx

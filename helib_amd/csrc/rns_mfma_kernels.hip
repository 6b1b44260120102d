// rns_mfma_kernels.hip -- the exact basis extension from 17..40 source primes with its target sums on the matrix cores
// (gfx950, V_MFMA_I32_32X32X32_I8): same contract, same words as rns_extend_wide_kernel (rns_kernels.h), which stays
// as the control (HX_NO_MFMA_EXT=1).  The reference: addPrimes / scaleDownToSet / breakIntoDigits at the chain of its
// own benchmark parameter (src/DoubleCRT.cpp:565-599, :1464-1516, :479-561; benchmarks/bgv_basic.cpp:247, bits = 6400:
// digits and dropped sets of 36 primes, up to 107 targets).  Method, table layout and the CPU restatement: mfma_ext.h.
//
// One wavefront = 64 coefficients, no workgroup-level cooperation (no LDS, no barrier):
//   1. front end, lane = coefficient (as the wide kernel, word for word): y_k = x_k (P/p_k)^-1 mod p_k, the quotient
//      cnt and the sign from the double-precision sum of y_k / p_k, the plaintext-space correction, value / P;
//      untrusted lanes go onto the redo list (the Garner pass behind this launch does them) and write nothing.
//   2. the y_k as packed balanced limbs ARE the B operand: lane l holds its coefficient's 8 bytes per source; one
//      v_permlane32_swap per register pair hands lanes l and l + 32 each other's half of the sources, which leaves
//      two 32-column operands (coefficients 0..31 and 32..63 of the wavefront) in MFMA layout.
//   3. per tile of four targets: 2 x steps MFMAs (K = 32 each: four source slots x 8 limbs per lane half) from the
//      plan's A table, accumulators started at base + delta so that every limb sum is a non-negative 24-bit number.
//   4. lane (col, h) then owns all eight limb sums of targets 4 tile + 2 h + {0, 1} for columns col and 32 + col:
//      recombine (80 bits), reduce modulo t (2^64 mod t by a 32-bit Shoup product, the low word by the 32-bit
//      reciprocal), correction, store / in-place update -- about 60 vector instructions per (coefficient, target)
//      where the wide kernel issues 144 multiply-adds at n = 36.
#include "dev_common.h"
#include "rns_types.h"
#include "mfma_ext.h"
#include "rns_mfma_dev.h"
#include "prof.h"

namespace hx {

typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));

constexpr int MFX_THREADS = 256;

// any 64-bit x -> [0,q), q > 2^32 (rns_kernels.h norm_any, restated: that header is engine.hip's alone)
__device__ __forceinline__ uint64_t mfx_norm(uint64_t x, uint64_t q, uint32_t mu32)
{
  const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
  const uint64_t t = (uint64_t)xh * mu32 + __umulhi(xl, mu32);
  const uint32_t e = (uint32_t)(t >> 32);
  const uint64_t nq = 0 - q;
  uint64_t r = (uint64_t)e * (uint32_t)nq + x;
  r += (uint64_t)(e * (uint32_t)(nq >> 32)) << 32;
  return csub(r, q);
}
// x c mod q into [0, 2q) for x < 2^32: wp32 = floor(c 2^32 / q)
__device__ __forceinline__ uint64_t mfx_shoup32(uint32_t x, uint64_t c, uint32_t wp32, uint64_t q)
{
  const uint32_t qh = __umulhi(x, wp32);
  return (uint64_t)x * c - (uint64_t)qh * q;
}

__device__ __forceinline__ void redo_append_mfx(uint32_t* redo, size_t i)   // (rns_kernels.h redo_append)
{
  const uint32_t slot = atomicAdd(&redo[0], 1u);
  redo[1 + slot] = (uint32_t)i;
}

struct MfxCoef {      // what the back end needs of one coefficient (exchanged between lanes l and l + 32)
  uint32_t flags;     // 1: trusted and inside the row; 2: dm_nonzero; 4: dm_negative
  uint32_t dm_lo, dm_hi;
};

template <int NSTEP>
__global__ void __launch_bounds__(MFX_THREADS, 2)
rns_extend_mfma_kernel(ExtPlanDev P, ExtArgs A, size_t row_words)
{
  constexpr int KS = 4 * NSTEP;          // source slots; the last one carries cnt
  const unsigned lane = threadIdx.x & 63u, h = lane >> 5, col = lane & 31u;
  const size_t wbase = (size_t)blockIdx.x * MFX_THREADS + (threadIdx.x & ~63u);
  const size_t i = wbase + lane;
  const bool live = i < row_words;
  const size_t il = live ? i : 0;        // (dead lanes read coefficient 0 and write nothing)
  const int n = P.n;

  // ---- 1. front end: rns_extend_wide_kernel's, lane = coefficient ----
  // (all source words first, then the arithmetic: one wait chain instead of a full memory round trip per source;
  // slots past n re-read the last source's word -- a cache hit -- so that the loads need no branch)
  uint64_t xs[KS - 1];
  static_for<0, KS - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    const int kk = k < n ? k : n - 1;
    xs[k] = ld_stream1(A.src + (size_t)A.src_row[kk] * row_words + il);
  });
  uint32_t yl[KS], yh[KS];
  double z = 0;
  const uint64_t p = P.ptxt;
  uint64_t pacc = 0;
  static_for<0, KS - 1>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    uint64_t packed = 0;
    if (k < n) {
      const uint64_t x = xs[k];
      if (A.own_dst_row[k] != 0xffff && live)
        st_stream1(A.dst + (size_t)A.own_dst_row[k] * row_words + i, x);
      const uint64_t pk = P.src_q[k];
      uint64_t y = shoup4(x, ld_tw(P.hps_inv, k), 0 - pk);  // any 64-bit x -> [0, 4 p_k)
      y = csub(y, pk + pk);
      y = csub(y, pk);
      z += (double)y * P.src_rq[k];
      if (p > 1) {
        pacc += shoup_lazy(y, ld_tw(P.Wp_hps, k), p);  // each < 2p
        if ((k & 3) == 3)
          pacc = red64(pacc, p, P.ptxt_mu64);
      }
      packed = mfx::pack_balanced(y);
    }
    uint32_t lo = (uint32_t)packed, hi = (uint32_t)(packed >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    yl[k] = lo;
    yh[k] = hi;
  });
  const double fl = floor(z), f = z - fl, eps = P.hps_eps;
  const bool trusted = !(f < eps || f > 1.0 - eps || fabs(f - 0.5) < eps);
  if (!trusted && live)
    redo_append_mfx(A.redo, i);
  const bool neg = f > 0.5;
  const uint32_t cnt = trusted ? (uint32_t)fl + (neg ? 1u : 0u) : 0u;   // <= n + 1 < 128: one non-negative limb
  yl[KS - 1] = cnt;
  yh[KS - 1] = 0;

  // BGV: make delta divisible by ptxtSpace (src/DoubleCRT.cpp:1485-1508)
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
  if (A.frac && trusted && live) {
    double fr = f - (neg ? 1.0 : 0.0);
    if (dm_nonzero)
      fr += dm_negative ? (double)dm_abs : -(double)dm_abs;
    A.frac[i] = fr;
  }

  // ---- 2. operand B: swap halves between lanes l and l + 32 ----
  // before: lane l holds all slots of coefficient l.  after: X registers (slots 4j, 4j+1) and Y registers (4j+2, 4j+3)
  // are the operands of columns 0..31 and 32..63: X upper lanes <- Y of the lower lanes, Y lower lanes <- X of the upper.
  mf_v4i B0[NSTEP], B1[NSTEP];
  static_for<0, NSTEP>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    uint32_t x[4] = {yl[4 * j], yh[4 * j], yl[4 * j + 1], yh[4 * j + 1]};
    uint32_t y[4] = {yl[4 * j + 2], yh[4 * j + 2], yl[4 * j + 3], yh[4 * j + 3]};
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const auto r = __builtin_amdgcn_permlane32_swap(x[d], y[d], false, false);
      x[d] = r[0];
      y[d] = r[1];
    }
    B0[j] = mf_v4i{(int)x[0], (int)x[1], (int)x[2], (int)x[3]};
    B1[j] = mf_v4i{(int)y[0], (int)y[1], (int)y[2], (int)y[3]};
  });
  MfxCoef C0, C1;   // of columns col and 32 + col
  {
    const uint32_t fl_ = ((trusted && live) ? 1u : 0u) | (dm_nonzero ? 2u : 0u) | (dm_negative ? 4u : 0u);
    uint32_t a[3] = {fl_, (uint32_t)dm_abs, (uint32_t)(dm_abs >> 32)}, b[3] = {a[0], a[1], a[2]};
#pragma unroll
    for (int d = 0; d < 3; d++) {
      const auto r = __builtin_amdgcn_permlane32_swap(a[d], b[d], false, false);
      a[d] = r[0];
      b[d] = r[1];
    }
    C0 = MfxCoef{a[0], a[1], a[2]};
    C1 = MfxCoef{b[0], b[1], b[2]};
  }

  // ---- 3. + 4. tiles of four targets ----
  const int nt = P.nt, ntile = (nt + 3) >> 2, stride = wide_stride(n);
  const mf_v4i* __restrict__ At = reinterpret_cast<const mf_v4i*>(P.mfma_a) + lane;
  const mf_v4i* __restrict__ It = reinterpret_cast<const mf_v4i*>(P.mfma_init) + 4 * h;
  const size_t i0 = wbase + col, i1 = wbase + 32 + col;   // the two coefficients this lane finishes
  for (int tau = 0; tau < ntile; tau++) {
    mf_v16i acc0, acc1;
    {
      mf_v16i init;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const mf_v4i v = It[(size_t)tau * 8 + g];
        init[4 * g] = v.x;
        init[4 * g + 1] = v.y;
        init[4 * g + 2] = v.z;
        init[4 * g + 3] = v.w;
      }
      const mf_v4i a0 = At[((size_t)tau * NSTEP) * 64];
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, B0[0], init, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, B1[0], init, 0, 0, 0);
    }
    static_for<1, NSTEP>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const mf_v4i a = At[((size_t)tau * NSTEP + j) * 64];
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B0[j], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, B1[j], acc1, 0, 0, 0);
    });
    static_for<0, 2>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const int t0 = 4 * tau + s, t1 = t0 + 2;          // (uniform) the targets of the lower / upper lanes
      const int t = h ? t1 : t0;
      if (t < nt) {
        // header of the wide kernel's record, per lane (two distinct addresses per wavefront)
        const uint64_t* rec = (const uint64_t*)(uintptr_t)P.wide_pack + (size_t)t * (size_t)stride;
        const uint64_t q = rec[0], pmod = rec[1], c64 = rec[2], mu64 = rec[4];
        const uint32_t wp32 = (uint32_t)(rec[3] >> 32), mu32 = (uint32_t)mu64;
        const int tc0 = t0 < nt ? t0 : nt - 1, tc1 = t1 < nt ? t1 : nt - 1;
        const uint32_t drow = h ? A.dst_row[tc1] : A.dst_row[tc0];
        const uint32_t urow = h ? A.upd_row[tc1] : A.upd_row[tc0];
        static_for<0, 2>([&](auto cbc) {
          constexpr int cb = decltype(cbc)::value;
          const MfxCoef& C = cb ? C1 : C0;
          if (C.flags & 1u) {
            const mf_v16i& acc = cb ? acc1 : acc0;
            const uint32_t S[8] = {(uint32_t)acc[8 * s], (uint32_t)acc[8 * s + 1], (uint32_t)acc[8 * s + 2], (uint32_t)acc[8 * s + 3],
                                   (uint32_t)acc[8 * s + 4], (uint32_t)acc[8 * s + 5], (uint32_t)acc[8 * s + 6], (uint32_t)acc[8 * s + 7]};
            const mfx::V80 v = mfx::recombine(S);
            uint64_t r = mfx_shoup32(v.hi, c64, wp32, q) + mfx_norm(v.lo, q, mu32);   // [0, 3q)
            r = csub(r, q + q);
            r = csub(r, q);
            if (C.flags & 2u) {
              // delta -= diffProd * delta_i_modP
              const uint64_t dm = ((uint64_t)C.dm_hi << 32) | C.dm_lo;
              uint64_t corr = dm;
              if (!P.corr_unit)
                corr = mul_shoup(red64(dm, q, mu64), pmod, rec[7], q);
              r = (C.flags & 4u) ? add_mod(r, corr, q) : sub_mod(r, corr, q);
            }
            const size_t ic = cb ? i1 : i0;
            if (drow != 0xffff)
              st_stream1(A.dst + (size_t)drow * row_words + ic, r);
            if (urow != 0xffff) {
              uint64_t* u = A.upd + (size_t)urow * row_words + ic;
              *u = mul_shoup(sub_mod(*u, r, q), rec[5], rec[6], q);
            }
          }
        });
      }
    });
  }
}

template <int NSTEP>
static hipError_t launch_mfx(const ExtPlanDev& P, const ExtArgs& A, size_t row_words, hipStream_t st)
{
  const dim3 grid((unsigned)((row_words + MFX_THREADS - 1) / MFX_THREADS)), block(MFX_THREADS);
  HX_LAUNCH((rns_extend_mfma_kernel<NSTEP>), grid, block, 0, st, P, A, row_words);
  return hipGetLastError();
}

hipError_t launch_rns_extend_mfma(const ExtPlanDev& P, const ExtArgs& A, size_t row_words, hipStream_t st)
{
  switch ((int)P.mfma_steps) {
    case 5: return launch_mfx<5>(P, A, row_words, st);
    case 6: return launch_mfx<6>(P, A, row_words, st);
    case 7: return launch_mfx<7>(P, A, row_words, st);
    case 8: return launch_mfx<8>(P, A, row_words, st);
    case 9: return launch_mfx<9>(P, A, row_words, st);
    case 10: return launch_mfx<10>(P, A, row_words, st);
    case 11: return launch_mfx<11>(P, A, row_words, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace hx

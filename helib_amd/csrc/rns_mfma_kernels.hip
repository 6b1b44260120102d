// rns_mfma_kernels.hip -- the exact basis extension from many source primes with its target sums on the matrix cores
// (gfx950, V_MFMA_I32_32X32X32_I8): same contract, same words as the VALU kernels it stands in for -- for 17 .. 40
// sources rns_extend_wide_kernel, for the 9 .. 16-source plans of the fast kernels rns_extend_fast_kernel<n, HPS>
// (rns_kernels.h), which stay as the controls (HX_NO_MFMA_EXT=1; HX_MFMA_MIN_N moves the lower end, instantiations
// exist from four sources on).  The reference: addPrimes / scaleDownToSet / breakIntoDigits
// (src/DoubleCRT.cpp:565-599, :1464-1516, :479-561) at the chain of its own benchmark parameter
// (benchmarks/bgv_basic.cpp:247, bits = 6400: digits and dropped sets of 36 primes, up to 107 targets) and the
// several-primes mod-switch of the CKKS chain one level down (8 + 3 dropped primes).  Method, table layout and the
// CPU restatement: mfma_ext.h.
//
// One wavefront = 32 coefficients, lanes l and l + 32 on the same one:
//   1. front end as the wide kernel's, split by source between the two lanes: y_k = x_k (P/p_k)^-1 mod p_k for the
//      lane's half of the sources, partial sums of y_k / p_k (double) and of the plaintext-space correction exchanged
//      once (v_permlane32_swap); quotient cnt, sign, correction, value / P then as there; untrusted coefficients go
//      onto the redo list (the Garner pass behind this launch does them) and write nothing.
//   2. the y_k as packed balanced limbs ARE the lane's B operand: 16 bytes (two sources) per MFMA step.
//   3. per tile of four targets: `steps` MFMAs (K = 32 each) from the plan's A table, staged through the LDS one tile
//      ahead together with the targets' constants; accumulators started at base + delta so that every limb sum is a
//      non-negative 24-bit number.
//   4. lane (col, h) then owns all eight limb sums of targets 4 tile + 2 h + {0, 1} of its coefficient: recombine
//      (80 bits), reduce modulo t (one 32-bit quotient estimate for t >= 2^48; else 2^64 mod t by a 32-bit Shoup
//      product and the low word by the 32-bit reciprocal), correction, store / in-place update -- about 45 vector
//      instructions per (coefficient, target) where the wide kernel issues 144 multiply-adds at n = 36.
#include "dev_common.h"
#include "rns_types.h"
#include "mfma_ext.h"
#include "rns_mfma_dev.h"
#include "prof.h"

namespace hx {

typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));

constexpr int MFX_THREADS = 256;   // four wavefronts of 32 coefficients
#ifndef MFX_MINW
#define MFX_MINW 4
#endif

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

// the value lane l ^ 32 holds
__device__ __forceinline__ uint32_t mfx_partner(uint32_t v, unsigned h)
{
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);   // r[0]: upper lanes <- lower; r[1]: lower <- upper
  return h ? r[0] : r[1];
}
__device__ __forceinline__ uint64_t mfx_partner64(uint64_t v, unsigned h)
{
  return ((uint64_t)mfx_partner((uint32_t)(v >> 32), h) << 32) | mfx_partner((uint32_t)v, h);
}

// per source prime, in the LDS (lanes l and l + 32 work on different sources: two addresses per wavefront)
struct MfxSrc {
  uint64_t q, hinv_w, hinv_wp, rq /* double bits */, wp_w, wp_wp;
  uint32_t row, own_row;
  uint64_t pad;
};
static_assert(sizeof(MfxSrc) == 64, "one source record = four 16-byte LDS reads");

// One wavefront = 32 coefficients: lanes l and l + 32 share coefficient l & 31.  In the MFMA operand layout lane half h
// supplies source slots 4 j + 2 h, 4 j + 2 h + 1 of step j and receives the limb sums of targets 4 tile + 2 h + {0, 1}
// -- so each half runs the front end for ITS sources only and the back end for ITS targets only, and the two exchange
// nothing but the partial sums that decide quotient, sign and correction (the first form of this kernel gave a
// wavefront 64 coefficients and two column blocks: 170 registers, two wavefronts per SIMD, 43 % of their cycles
// waiting: 147 us per launch at n = 36 onto 107 targets where its instructions needed 50).
template <int NSTEP>
__global__ void __launch_bounds__(MFX_THREADS, MFX_MINW)
rns_extend_mfma_kernel(ExtPlanDev P, ExtArgs A, size_t row_words)
{
  constexpr int KL = 2 * NSTEP;          // source slots of one lane half; the last one of half 1 carries cnt
  const unsigned lane = threadIdx.x & 63u, h = lane >> 5, col = lane & 31u;
  const size_t i = (size_t)blockIdx.x * (MFX_THREADS / 2) + (threadIdx.x >> 6) * 32 + col;
  const bool live = i < row_words;
  const size_t il = live ? i : 0;        // (dead lanes read coefficient 0 and write nothing)
  const int n = P.n;
  const int nt = P.nt, ntile = (nt + 3) >> 2, stride = wide_stride(n);
  const uint32_t rw32 = (uint32_t)row_words;   // (the launch is for row_words < 2^32)

  constexpr int TV = NSTEP * 64 + mfx::EXTRA_VECS;             // 16-byte vectors of one tile block
  constexpr int NLD = (TV + MFX_THREADS - 1) / MFX_THREADS;    // ... per thread
  __shared__ mf_v4i a_lds[2 * TV];
  __shared__ uint32_t rows_lds[MAX_ROWS + 4];   // dst_row | upd_row << 16 per target
  __shared__ MfxSrc src_lds[4 * mfx::MAX_STEPS];
  const mf_v4i* __restrict__ Ag = reinterpret_cast<const mf_v4i*>(P.mfma_a);
  mf_v4i stage[NLD];
  // (no branch around a load: out-of-range threads re-read the block's last vector and drop it)
  auto fetch = [&](int tau) {
    static_for<0, NLD>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      const unsigned idx = threadIdx.x + MFX_THREADS * l;
      stage[l] = Ag[(size_t)tau * TV + (idx < (unsigned)TV ? idx : (unsigned)TV - 1u)];
    });
  };
  auto put = [&](int buf) {
    static_for<0, NLD>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      const unsigned idx = threadIdx.x + MFX_THREADS * l;
      if (NLD * MFX_THREADS == TV || idx < (unsigned)TV)
        a_lds[buf * TV + idx] = stage[l];
    });
  };
  // the coefficient's source words, requested before anything else (local slot e of half h = source
  // 4 (e >> 1) + 2 h + (e & 1); slots past n re-read the last source's word -- a cache hit -- and multiply it by nothing)
  uint64_t xs[KL];
  static_for<0, KL>([&](auto ec) {
    constexpr int e = decltype(ec)::value;
    constexpr int k0 = 4 * (e >> 1) + (e & 1), k1 = k0 + 2;   // of the lower / upper lanes
    const uint32_t row = h ? A.src_row[k1 < n ? k1 : n - 1] : A.src_row[k0 < n ? k0 : n - 1];
    xs[e] = ld_stream1(A.src + ((uint64_t)row * rw32 + il));
  });
  fetch(0);
  // ---- 0. the launch's tables into the LDS ----
  for (int t = (int)threadIdx.x; t < ((nt + 3) & ~3); t += MFX_THREADS)
    rows_lds[t] = t < nt ? ((uint32_t)A.dst_row[t] | ((uint32_t)A.upd_row[t] << 16)) : 0xffffffffu;
  const uint64_t p = P.ptxt;
  if ((int)threadIdx.x < 4 * NSTEP) {
    const int k = (int)threadIdx.x, kk = k < n ? k : n - 1;   // (slots past n: the last source's word, multiplied by nothing)
    MfxSrc r;
    r.q = P.src_q[kk];
    const TW hi = ld_tw(P.hps_inv, kk);
    r.hinv_w = hi.w;
    r.hinv_wp = hi.wp;
    r.rq = __double_as_longlong(P.src_rq[kk]);
    r.wp_w = r.wp_wp = 0;
    if (p > 1) {
      const TW wp = ld_tw(P.Wp_hps, kk);
      r.wp_w = wp.w;
      r.wp_wp = wp.wp;
    }
    r.row = A.src_row[kk];
    r.own_row = A.own_dst_row[kk];
    r.pad = 0;
    src_lds[k] = r;
  }
  put(0);
  __syncthreads();

  // ---- 1. front end (rns_extend_wide_kernel's, split between the two lanes of a coefficient by source) ----
  uint32_t yl[KL], yh[KL];
  double z = 0;
  uint64_t pacc = 0;
  static_for<0, KL>([&](auto ec) {
    constexpr int e = decltype(ec)::value;
    const int k = 4 * (e >> 1) + 2 * (int)h + (e & 1);
    const MfxSrc& S = src_lds[k];
    const uint64_t x = xs[e];
    const uint32_t own = S.own_row;
    if (own != 0xffffu && live && k < n)
      st_stream1(A.dst + ((uint64_t)own * rw32 + i), x);
    const uint64_t pk = S.q;
    TW hinv;
    hinv.w = S.hinv_w;
    hinv.wp = S.hinv_wp;
    uint64_t y = shoup4(x, hinv, 0 - pk);  // any 64-bit x -> [0, 4 p_k)
    y = csub(y, pk + pk);
    y = csub(y, pk);
    y = k < n ? y : 0;
    z += (double)y * __longlong_as_double(S.rq);
    if (p > 1) {
      TW wp;
      wp.w = S.wp_w;
      wp.wp = S.wp_wp;
      pacc += shoup_lazy(y, wp, p);  // each < 2p
      if ((e & 3) == 3)
        pacc = red64(pacc, p, P.ptxt_mu64);
    }
    const uint64_t packed = mfx::pack_balanced(y);
    uint32_t lo = (uint32_t)packed, hi = (uint32_t)(packed >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    yl[e] = lo;
    yh[e] = hi;
  });
  // both lanes of a coefficient form the same sum in the same order: half 0's part + half 1's
  {
    const double zo = __longlong_as_double(mfx_partner64(__double_as_longlong(z), h));
    z = h ? zo + z : z + zo;
  }
  const double fl = floor(z), f = z - fl, eps = P.hps_eps;
  const bool trusted = !(f < eps || f > 1.0 - eps || fabs(f - 0.5) < eps);
  if (!trusted && live && h == 0)
    redo_append_mfx(A.redo, i);
  const bool neg = f > 0.5;
  const uint32_t cnt = trusted ? (uint32_t)fl + (neg ? 1u : 0u) : 0u;   // <= n + 1 < 128: one non-negative limb
  if (h) {   // slot 4 NSTEP - 1: the last one of half 1 (beyond n by construction)
    yl[KL - 1] = cnt;
    yh[KL - 1] = 0;
  }
  // BGV: make delta divisible by ptxtSpace (src/DoubleCRT.cpp:1485-1508)
  bool dm_nonzero = false, dm_negative = false;
  uint64_t dm_abs = 0;
  if (p > 1) {
    uint64_t r = red64(pacc, p, P.ptxt_mu64);
    r = add_mod(r, mfx_partner64(r, h), p);
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
  const bool ok = trusted && live;
  if (A.frac && ok && h == 0) {
    double fr = f - (neg ? 1.0 : 0.0);
    if (dm_nonzero)
      fr += dm_negative ? (double)dm_abs : -(double)dm_abs;
    A.frac[i] = fr;
  }
  // ---- 2. operand B: each lane's own packed limbs, two sources (16 bytes) per step ----
  mf_v4i B[NSTEP];
  static_for<0, NSTEP>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    B[j] = mf_v4i{(int)yl[2 * j], (int)yh[2 * j], (int)yl[2 * j + 1], (int)yh[2 * j + 1]};
  });

  // ---- 3. + 4. tiles of four targets ----
  // Nothing in this loop waits for global memory it has just asked for: a tile's block -- the A operand (steps x 1 KB)
  // and its targets' constants (accumulator starts, t, floor(2^80 / t), P^-1 mod t: mfma_ext.h) -- goes through the
  // LDS, double-buffered: the workgroup's wavefronts need the same block, so each thread fetches its share of the
  // NEXT tile's (issued before this tile's MFMAs, stored behind them and before this tile's own stores -- one counter
  // serves loads and stores here -- one barrier per tile); the output / update rows of the launch sit in the LDS from
  // the start; the words an in-place update reads are requested before the MFMAs.
  // (Round-6 record of the 64-coefficient form, n = 36 onto 107 targets, 2^18 coefficients: every wavefront loading its
  // own operands one step ahead 303 us, a whole tile ahead in registers 252 us, through the LDS but constants and
  // update words still loaded where they are used 250 us, as described here 147 us.)
  const uint64_t* wide = (const uint64_t*)(uintptr_t)P.wide_pack;
  const uint64_t* const safe = A.src + (size_t)A.src_row[0] * row_words;   // what a lane without an update row reads instead
  for (int tau = 0; tau < ntile; tau++) {
    mf_v16i acc;
    fetch(tau + 1 < ntile ? tau + 1 : tau);   // (the last tile re-reads itself)
    const mf_v4i* ac = a_lds + (tau & 1) * TV + lane;
    const mf_v4i* ex = a_lds + (tau & 1) * TV + NSTEP * 64 + 8 * h;   // this lane half's constants
    // rows of this lane's two targets; the words their in-place updates will read, requested now
    const uint32_t rw0 = rows_lds[4 * tau + 2 * h], rw1 = rows_lds[4 * tau + 2 * h + 1];
    uint64_t u0 = 0, u1 = 0;
    if (4 * tau < A.nu) {   // (uniform: the targets with an update row come first, ExtArgs::nu of them)
      u0 = *(((rw0 >> 16) != 0xffffu && ok) ? A.upd + ((uint64_t)(rw0 >> 16) * rw32 + i) : safe);
      u1 = *(((rw1 >> 16) != 0xffffu && ok) ? A.upd + ((uint64_t)(rw1 >> 16) * rw32 + i) : safe);
    }
    __builtin_amdgcn_sched_barrier(0);   // (the loads above are issued here, not where their values are used)
    {
      mf_v16i init;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const mf_v4i v = ex[g];
        init[4 * g] = v.x;
        init[4 * g + 1] = v.y;
        init[4 * g + 2] = v.z;
        init[4 * g + 3] = v.w;
      }
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ac[0], B[0], init, 0, 0, 0);
      static_for<1, NSTEP>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ac[j * 64], B[j], acc, 0, 0, 0);
      });
    }
    // the next tile's block into the other buffer now -- before this tile's stores are issued, so that the wait for the
    // staged vectors does not wait for those stores as well (one counter for both on this target)
    put((tau + 1) & 1);
    const mf_v4i mu80s = ex[mfx::EX_MU80 / 4], qs = ex[mfx::EX_Q / 4];
    static_for<0, 2>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const int t = 4 * tau + 2 * (int)h + s;
      if (t < nt && ok) {
        const uint64_t q = s ? (((uint64_t)(uint32_t)qs.w << 32) | (uint32_t)qs.z) : (((uint64_t)(uint32_t)qs.y << 32) | (uint32_t)qs.x);
        const uint32_t mu80 = (uint32_t)(s ? mu80s.y : mu80s.x);
        const uint32_t rws = s ? rw1 : rw0, drow = rws & 0xffffu, urow = rws >> 16;
        const uint64_t* rec = wide + (uint32_t)t * (uint32_t)stride;   // (the wide kernel's record: the rarer constants)
        const uint32_t S[8] = {(uint32_t)acc[8 * s], (uint32_t)acc[8 * s + 1], (uint32_t)acc[8 * s + 2], (uint32_t)acc[8 * s + 3],
                               (uint32_t)acc[8 * s + 4], (uint32_t)acc[8 * s + 5], (uint32_t)acc[8 * s + 6], (uint32_t)acc[8 * s + 7]};
        const mfx::V80 v = mfx::recombine(S);
        uint64_t r;
        if (mu80) {   // t >= 2^48 (every prime of the benchmark chains): one 32-bit quotient estimate, [0, 4t)
          r = mfx::red80_lazy(v.lo, v.hi, q, mu80);
        } else {      // any t > 2^32: 2^64 mod t by a 32-bit Shoup product, the low word by the 32-bit reciprocal
          const uint64_t c64 = rec[2];
          const uint32_t wp32 = (uint32_t)(rec[3] >> 32), mu32 = (uint32_t)rec[4];
          r = mfx_shoup32(v.hi, c64, wp32, q) + mfx_norm(v.lo, q, mu32);   // [0, 3t)
        }
        r = csub(r, q + q);
        r = csub(r, q);
        if (dm_nonzero) {
          // delta -= diffProd * delta_i_modP
          uint64_t corr = dm_abs;
          if (!P.corr_unit)
            corr = mul_shoup(red64(dm_abs, q, rec[4]), rec[1], rec[7], q);
          r = dm_negative ? add_mod(r, corr, q) : sub_mod(r, corr, q);
        }
        if (drow != 0xffffu)
          st_stream1(A.dst + ((uint64_t)drow * rw32 + i), r);
        if (urow != 0xffffu) {
          const uint64_t uold = s ? u1 : u0;
          const mf_v4i uc = ex[mfx::EX_UPD / 4 + s];   // P^-1 mod t, its Shoup companion
          const uint64_t uw = ((uint64_t)(uint32_t)uc.y << 32) | (uint32_t)uc.x, uwp = ((uint64_t)(uint32_t)uc.w << 32) | (uint32_t)uc.z;
          A.upd[(uint64_t)urow * rw32 + i] = mul_shoup(sub_mod(uold, r, q), uw, uwp, q);
        }
      }
    });
    __syncthreads();
  }
}

template <int NSTEP>
static hipError_t launch_mfx(const ExtPlanDev& P, const ExtArgs& A, size_t row_words, hipStream_t st)
{
  const dim3 grid((unsigned)((row_words + MFX_THREADS / 2 - 1) / (MFX_THREADS / 2))), block(MFX_THREADS);
  HX_LAUNCH((rns_extend_mfma_kernel<NSTEP>), grid, block, 0, st, P, A, row_words);
  return hipGetLastError();
}

hipError_t launch_rns_extend_mfma(const ExtPlanDev& P, const ExtArgs& A, size_t row_words, hipStream_t st)
{
  switch ((int)P.mfma_steps) {
    case 2: return launch_mfx<2>(P, A, row_words, st);
    case 3: return launch_mfx<3>(P, A, row_words, st);
    case 4: return launch_mfx<4>(P, A, row_words, st);
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

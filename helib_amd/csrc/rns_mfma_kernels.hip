// rns_mfma_kernels.hip -- the exact basis extension from 17..40 source primes with its target sums on the matrix cores
// (gfx950, V_MFMA_I32_32X32X32_I8): same contract, same words as rns_extend_wide_kernel (rns_kernels.h), which stays
// as the control (HX_NO_MFMA_EXT=1).  The reference: addPrimes / scaleDownToSet / breakIntoDigits at the chain of its
// own benchmark parameter (src/DoubleCRT.cpp:565-599, :1464-1516, :479-561; benchmarks/bgv_basic.cpp:247, bits = 6400:
// digits and dropped sets of 36 primes, up to 107 targets).  Method, table layout and the CPU restatement: mfma_ext.h.
//
// One wavefront = 64 coefficients; the four wavefronts of a workgroup share only the table reads (step 3):
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
  // Nothing in this loop waits for global memory it has just asked for: a tile's block -- the A operand (steps x 1 KB)
  // and its targets' constants (accumulator starts, t, floor(2^80 / t): mfma_ext.h) -- goes through the LDS,
  // double-buffered: the workgroup's four wavefronts need the same block, so each fetches a quarter of the NEXT tile's
  // (issued before this tile's MFMAs, stored behind its reductions, one barrier per tile); the output / update rows
  // of the launch sit in the LDS from the start; the words an in-place update reads are requested before the MFMAs.
  // (Round-6 record, n = 36 onto 107 targets, 2^18 coefficients: every wavefront loading its own operands one step
  // ahead 303 us, a whole tile ahead in registers 252 us, through the LDS but constants and update words still loaded
  // where they are used 250 us.)
  const int nt = P.nt, ntile = (nt + 3) >> 2, stride = wide_stride(n);
  const size_t i0 = wbase + col, i1 = wbase + 32 + col;   // the two coefficients this lane finishes
  const uint64_t* wide = (const uint64_t*)(uintptr_t)P.wide_pack;
  const uint32_t rw32 = (uint32_t)row_words;   // (the launch is for row_words < 2^32)
  constexpr int TV = NSTEP * 64 + mfx::EXTRA_VECS;             // 16-byte vectors of one tile block
  constexpr int NLD = (TV + MFX_THREADS - 1) / MFX_THREADS;    // ... per thread
#ifdef MFX_LDS_PAD_KB   // (A/B probe: more LDS per workgroup = fewer resident workgroups)
  __shared__ mf_v4i a_lds[2 * TV + MFX_LDS_PAD_KB * 64];
#else
  __shared__ mf_v4i a_lds[2 * TV];
#endif
  __shared__ uint32_t rows_lds[MAX_ROWS + 4];   // dst_row | upd_row << 16 per target
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
  const uint64_t* const safe = A.src + (size_t)A.src_row[0] * row_words;   // what a lane without an update row reads instead
#ifdef MFX_NO_LDS   // (A/B probe: every wavefront keeps the next tile's block in registers -- no sharing, no barrier)
  mf_v4i an[NSTEP], exn[6];
  auto fetchw = [&](int tau) {
    static_for<0, NSTEP>([&](auto jc) { an[decltype(jc)::value] = Ag[(size_t)tau * TV + decltype(jc)::value * 64 + lane]; });
    static_for<0, 6>([&](auto gc) { exn[decltype(gc)::value] = Ag[(size_t)tau * TV + NSTEP * 64 + 8 * h + decltype(gc)::value]; });
  };
  fetchw(0);
#else
  fetch(0);
#endif
  for (int t = (int)threadIdx.x; t < ((nt + 3) & ~3); t += MFX_THREADS)
    rows_lds[t] = t < nt ? ((uint32_t)A.dst_row[t] | ((uint32_t)A.upd_row[t] << 16)) : 0xffffffffu;
#ifndef MFX_NO_LDS
  put(0);
#endif
  __syncthreads();
  for (int tau = 0; tau < ntile; tau++) {
    mf_v16i acc0, acc1;
#ifdef MFX_NO_LDS
    mf_v4i acr[NSTEP], ex[6];
    static_for<0, NSTEP>([&](auto jc) { acr[decltype(jc)::value] = an[decltype(jc)::value]; });
    static_for<0, 6>([&](auto gc) { ex[decltype(gc)::value] = exn[decltype(gc)::value]; });
    fetchw(tau + 1 < ntile ? tau + 1 : tau);
#else
    fetch(tau + 1 < ntile ? tau + 1 : tau);   // (the last tile re-reads itself)
    const mf_v4i* ac = a_lds + (tau & 1) * TV + lane;
    const mf_v4i* ex = a_lds + (tau & 1) * TV + NSTEP * 64 + 8 * h;   // this lane half's constants
#endif
    // rows of this lane's two targets; the words their in-place updates will read (two coefficients each), requested now
    const uint32_t rw0 = rows_lds[4 * tau + 2 * h], rw1 = rows_lds[4 * tau + 2 * h + 1];
    uint64_t u00 = 0, u01 = 0, u10 = 0, u11 = 0;   // [s][cb]
    if (4 * tau < A.nu) {   // (uniform: the targets with an update row come first, ExtArgs::nu of them)
      const bool up0 = (rw0 >> 16) != 0xffffu, up1 = (rw1 >> 16) != 0xffffu, ok0 = C0.flags & 1u, ok1 = C1.flags & 1u;
      u00 = *((up0 && ok0) ? A.upd + ((uint64_t)(rw0 >> 16) * rw32 + i0) : safe);
      u01 = *((up0 && ok1) ? A.upd + ((uint64_t)(rw0 >> 16) * rw32 + i1) : safe);
      u10 = *((up1 && ok0) ? A.upd + ((uint64_t)(rw1 >> 16) * rw32 + i0) : safe);
      u11 = *((up1 && ok1) ? A.upd + ((uint64_t)(rw1 >> 16) * rw32 + i1) : safe);
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
      mf_v4i a[NSTEP];   // (all of the tile's operand first: the MFMAs then run back to back, not one LDS round trip apart)
#ifdef MFX_NO_LDS
      static_for<0, NSTEP>([&](auto jc) { a[decltype(jc)::value] = acr[decltype(jc)::value]; });
#else
      static_for<0, NSTEP>([&](auto jc) { a[decltype(jc)::value] = ac[decltype(jc)::value * 64]; });
#endif
      acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], B0[0], init, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], B1[0], init, 0, 0, 0);
#ifdef MFX_EXP_NOMFMA    // (timing probe: one MFMA step per tile instead of all)
      static_for<1, 1>([&](auto jc) {
#else
      static_for<1, NSTEP>([&](auto jc) {
#endif
        constexpr int j = decltype(jc)::value;
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j], B0[j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[j], B1[j], acc1, 0, 0, 0);
      });
    }
    // the next tile's block into the other buffer now -- before this tile's stores are issued, so that the wait for the
    // staged vectors does not wait for those stores as well (one counter for both on this target)
#ifndef MFX_NO_LDS
    put((tau + 1) & 1);
#endif
    const mf_v4i mu80s = ex[mfx::EX_MU80 / 4], qs = ex[mfx::EX_Q / 4], up01 = ex[mfx::EX_UPD / 4], up23 = ex[mfx::EX_UPD / 4 + 1];
    static_for<0, 2>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      const int t = 4 * tau + 2 * (int)h + s;
      if (t < nt) {
        const uint64_t q = s ? (((uint64_t)(uint32_t)qs.w << 32) | (uint32_t)qs.z) : (((uint64_t)(uint32_t)qs.y << 32) | (uint32_t)qs.x);
        const uint32_t mu80 = (uint32_t)(s ? mu80s.y : mu80s.x);
        const uint32_t rws = s ? rw1 : rw0, drow = rws & 0xffffu, urow = rws >> 16;
        const uint64_t* rec = wide + (uint32_t)t * (uint32_t)stride;   // (the wide kernel's record: the rarer constants)
        static_for<0, 2>([&](auto cbc) {
          constexpr int cb = decltype(cbc)::value;
          const MfxCoef& C = cb ? C1 : C0;
          if (C.flags & 1u) {
            const mf_v16i& acc = cb ? acc1 : acc0;
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
            if (C.flags & 2u) {
              // delta -= diffProd * delta_i_modP
              const uint64_t dm = ((uint64_t)C.dm_hi << 32) | C.dm_lo;
              uint64_t corr = dm;
              if (!P.corr_unit)
                corr = mul_shoup(red64(dm, q, rec[4]), rec[1], rec[7], q);
              r = (C.flags & 4u) ? add_mod(r, corr, q) : sub_mod(r, corr, q);
            }
            const uint64_t ic = cb ? i1 : i0;
#ifdef MFX_EXP_NOSTORE   // (timing probe: results dropped unless a value that never occurs shows up)
            if (drow != 0xffffu && r == 0xdeadbeefdeadbeefull)
#else
            if (drow != 0xffffu)
#endif
              st_stream1(A.dst + ((uint64_t)drow * rw32 + ic), r);
            if (urow != 0xffffu) {
              const uint64_t uold = s ? (cb ? u11 : u10) : (cb ? u01 : u00);
              const mf_v4i uc = s ? up23 : up01;   // P^-1 mod t, its Shoup companion
              const uint64_t uw = ((uint64_t)(uint32_t)uc.y << 32) | (uint32_t)uc.x, uwp = ((uint64_t)(uint32_t)uc.w << 32) | (uint32_t)uc.z;
              A.upd[(uint64_t)urow * rw32 + ic] = mul_shoup(sub_mod(uold, r, q), uw, uwp, q);
            }
          }
        });
      }
    });
#ifndef MFX_NO_LDS
    __syncthreads();
#endif
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

// conv_kernels.hip -- the fused convolution row kernel of the Bluestein path (general m) for gfx950:
// its own translation unit (twelve instantiations of a forward + inverse transform; builds in parallel
// with ntt_kernels.hip).  Descriptors: conv_dev.h; phase functions: ntt_core.h; callers: engine.hip
// (bluestein_rows).
#include "dev_common.h"
#include "conv_dev.h"
#include "ntt_kernel_util.h"
#include "prof.h"

namespace hx {

// ---------------------------------------------------------------------
// Convolution rows (general m): one workgroup = element source -> forward transform -> product with
// the precomputed transform of the fixed operand -> inverse transform -> store, in registers
// (conv_dev.h).  BluesteinFFT's  TofftRep_trunc / mul / FromfftRep  (src/bluestein.cpp:167-171, 189-193)
// and the two FFT multiplications of `rem Phi_m` (src/NumbTh.cpp:1741-1804) as ONE launch each.
// ---------------------------------------------------------------------
// a wave-uniform 64-bit value read through a plain pointer arrives in vector registers; this moves it
// to scalar ones (the convolution kernel holds ~30 such values across sixteen phases: in VGPRs they
// cost the register file more than the coefficient file can spare)
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ TW uniform_tw(const TW& t)
{
  TW r;
  r.w = uniform_u64(t.w);
  r.wp = uniform_u64(t.wp);
  return r;
}

template <uint32_t SRC, uint32_t DST>
struct ConvIO {
  static constexpr int LOAD_BOUND = 1;
  static constexpr bool LAZY_STORE = true;   // the forward side "stores" into the register file: pointwise product
  static constexpr bool PIPELINED = false;
  static constexpr bool SKIP_LOAD = true;    // ... and the inverse side starts from it
  // Proth-form rows: the pointwise product is ONE Montgomery product by the plain table word -- below 2q, carrying
  // 2^-64, which the inverse transform's last stage gives back (last_tw below)
  template <class AR>
  static constexpr int inv_load_bound() { return AR::PROTH ? 2 : 1; }
  // (Tried in round 5 and reverted, as round 4's lazy Shoup load was: the element source's own products -- pre-twist by
  // powers[i], split factor -- as lazy Montgomery products, a0 +- T a1 below 4q with no conditional subtraction.  Bit-exact,
  // and 5 x slower: 2.7 ms instead of 0.45 per 512 rows; the three products per element with their uniform table
  // pointers inside the 32-element load loop push the first register pass into scratch.  The load stays canonical.)
  uint64_t r2;              // 2^128 mod q (PrimeDev::r2)
  struct StorePrefetch {};
  const uint64_t* src;      // segment of the source
  uint64_t* dst;            // segment of the destination
  const uint64_t* aux;      // CONV_DST_FINAL: x (the dividend of rem Phi_m)
  const TW* hat;            // this unit's [Q] transform of the fixed operand
  const BluePrimeDev* P;    // q, powers / ipowers, m^-1: read where they are used (scalar registers, short live ranges)
  const SplitTW* S;         // split constants of this row
  const int32_t* zidx;
  uint32_t g;
  uint32_t split, Q, phim, m, d, base, alias;
  __device__ __forceinline__ uint64_t element(unsigned i, const TW* pw, uint64_t q) const
  {
    if constexpr (SRC == CONV_SRC_BLUE_PRE) {
      return i < phim ? shoup_full(src[i], pw[i], q) : 0;
    } else if constexpr (SRC == CONV_SRC_SCATTER) {
      if (i >= m)
        return 0;
      const int32_t j = zidx[i];
      return j >= 0 ? shoup_full(src[j], pw[i], q) : 0;
    } else {
      return i <= d ? src[base - i] : 0;
    }
  }
  __device__ __forceinline__ uint64_t load(unsigned tid, unsigned c) const
  {
    const unsigned p = tid + c;
    const uint64_t q = uniform_u64(P->q);
    const TW* pw = (const TW*)uniform_u64((uint64_t)(SRC == CONV_SRC_SCATTER ? P->ipowers : P->powers));
    const uint64_t a0 = element(p, pw, q);
    if (SRC == CONV_SRC_REV || split == 1)   // (the reversal sources are never split)
      return a0;
    // the source's support ends at phim (forward) or m (inverse): where a1 = 0 all four sub-blocks take a0 as it is.
    // Config 5 (m = 21845, phim = Q = 16384): the whole forward pass, and two thirds of the inverse one, skip the
    // second element and its product (the reference truncates the same way, src/bluestein.cpp:167, 189)
    if (p + Q >= (SRC == CONV_SRC_SCATTER ? m : phim))
      return a0;
    // radix-4 split of an input whose upper half is zero (every source here has support below 2Q):
    // b[g] = a0 +- T * a1 with T = T2 (g = 0, 1) or T3 (g = 2, 3)   (split_fwd4 with a2 = a3 = 0)
    const TW comb = uniform_tw(g < 2 ? S->T2 : S->T3);
    const uint64_t u = shoup_full(element(p + Q, pw, q), comb, q);
    return (g & 1u) ? subm(a0, u, q) : addm(a0, u, q);
  }
  template <int LOGN>
  __device__ __forceinline__ void store_prefetch(unsigned, StorePrefetch&) const {}
  // the forward transform's "store": v[i] <- v[i] * hat[i], canonical (the inverse transform's input)
  template <int LOGN, class AR, int B, bool EST>
  __device__ __forceinline__ void store_all(unsigned tid, uint64_t (&v)[32], const QC& qc, StorePrefetch&) const
  {
    // groups of two pairs (8 registers) ahead of their arithmetic: all 32 pairs at once would need 128
    // registers on top of the coefficient file
    constexpr int IOG = 2;
    static_for<0, 32 / IOG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      TW hb[IOG];
      static_for<0, IOG>([&](auto J) {
        constexpr int j = decltype(J)::value;
        hb[j] = hat[tid + eval_const<LOGN>(g * IOG + j)];
      });
      static_for<0, IOG>([&](auto J) {
        constexpr int j = decltype(J)::value, i = g * IOG + j;
        if constexpr (AR::PROTH) {
          uint64_t x = v[i];
          if constexpr (B > 12)
            x = csub(x, qc.q8);          // (the multiplied operand must stay below 12.9 q)
          v[i] = mont_mul(x, hb[j].w, qc);   // x hat 2^-64 in (0, 2q): not normalised, the inverse takes bound 2
        } else {
          // (shoup4 takes any 64-bit value: the lazy [0, Bq) output of the last pass as it is)
          v[i] = norm_from<4>(shoup4(v[i], hb[j], qc.nq), qc);
        }
      });
      HX_SCHED_FENCE();
    });
  }
  __device__ __forceinline__ void store(unsigned tid, unsigned c, uint64_t x) const
  {
    const unsigned p = tid + c;
    if constexpr (DST == CONV_DST_SUB) {
      dst[p] = x;
    } else {
      const uint64_t q = uniform_u64(P->q);
      const TW minv = uniform_tw(P->minv);
      if (p < phim) {
        // r = x - Q Phi_m; with the product modulo X^Q + 1 (alias) its coefficient p is (Q Phi)_p - x_(p + Q)
        uint64_t a = aux[p];
        if (alias && p + Q < m)
          a = subm(a, aux[p + Q], q);
        dst[p] = shoup_full(subm(a, x, q), minv, q);
      }
    }
  }
  __device__ __forceinline__ TW last_tw(TW def, int) const { return def; }
  // the last inverse stage's constants with one more 2^64: def = w 2^64 mod q, times 2^128 as a Montgomery product
  // = w 2^128 -- the Proth form of (w 2^64), which undoes the pointwise product's 2^-64
  __device__ __forceinline__ TWM last_tw(TWM def, int) const
  {
    const QC qc = make_qc(uniform_u64(P->q), 0);   // (q, qh, c1 are all this needs: no reciprocal)
    return csub(mont_mul(def, r2, qc), qc.q);
  }
};

template <int LOGN, class AR, class IO>
__device__ __forceinline__ void conv_body(uint32_t* lds, const IO& io, const typename AR::Tw* twf, const typename AR::Tw* twi,
                                          const QC& q)
{
  using RN = RowNTT<LOGN, AR>;
  uint64_t v[32];
  uint32_t nl[32];
  const unsigned w = wave_index();
  RN::template fwd<0>(fresh_tid(w), v, nl, lds, io, twf, q);
  __syncthreads();
  RN::template fwd<1>(fresh_tid(w), v, nl, lds, io, twf, q);
  __syncthreads();
  RN::template fwd<2>(fresh_tid(w), v, nl, lds, io, twf, q);
  __syncthreads();
  RN::template fwd<3>(fresh_tid(w), v, nl, lds, io, twf, q);
  __syncthreads();
  RN::template fwd<4>(fresh_tid(w), v, nl, lds, io, twf, q);
  __syncthreads();
  RN::template fwd<5>(fresh_tid(w), v, nl, lds, io, twf, q);
  __syncthreads();
  RN::template fwd<6>(fresh_tid(w), v, nl, lds, io, twf, q);
  __syncthreads();
  RN::template fwd<7>(fresh_tid(w), v, nl, lds, io, twf, q);   // ends with the pointwise product (ConvIO::store_all)
  __syncthreads();                                              // (the transposes' LDS is reused by the inverse)
  RN::template inv<0>(fresh_tid(w), v, nl, lds, io, twi, q);
  __syncthreads();
  RN::template inv<1>(fresh_tid(w), v, nl, lds, io, twi, q);
  __syncthreads();
  RN::template inv<2>(fresh_tid(w), v, nl, lds, io, twi, q);
  __syncthreads();
  RN::template inv<3>(fresh_tid(w), v, nl, lds, io, twi, q);
  __syncthreads();
  RN::template inv<4>(fresh_tid(w), v, nl, lds, io, twi, q);
  __syncthreads();
  RN::template inv<5>(fresh_tid(w), v, nl, lds, io, twi, q);
  __syncthreads();
  RN::template inv<6>(fresh_tid(w), v, nl, lds, io, twi, q);
  __syncthreads();
  RN::template inv<7>(fresh_tid(w), v, nl, lds, io, twi, q);
}

template <int LOGN, uint32_t SRC, uint32_t DST>
__global__ void __launch_bounds__(Geo<LOGN>::T, HX_NTT_MINWAVES(LOGN))
ntt_conv_kernel(ConvRowArgs A, ConvRows R, const PrimeDev* __restrict__ cprimes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned b = wid % A.batch, u = wid / A.batch, ri = u / A.split, g = u % A.split;
  const PrimeDev* pd = cprimes + uniform_u16(R.pd, u);
  const BluePrimeDev* P = R.bp[ri];
  constexpr unsigned Q = Geo<LOGN>::N;
  ConvIO<SRC, DST> io;
  const size_t seg = (size_t)ri * A.batch + b;
  const size_t polyseg = ((size_t)uniform_u16(R.row, ri) * A.batch + b) * A.phim;
  io.src = SRC == CONV_SRC_REV ? A.in + seg * A.in_stride : A.in + polyseg;
  io.dst = DST == CONV_DST_SUB ? A.out + ((size_t)u * A.batch + b) * Q : A.out + polyseg;
  io.aux = DST == CONV_DST_FINAL ? A.aux + seg * A.aux_stride : nullptr;
  io.hat = R.hat[ri] + (size_t)g * Q;
  io.P = P;
  io.S = &R.cp[ri]->S;
  io.zidx = A.zidx;
  io.g = g;
  io.split = A.split;
  io.Q = Q;
  io.phim = A.phim;
  io.m = A.m;
  io.d = A.d;
  io.base = A.base;
  io.alias = A.alias;
  io.r2 = pd->r2;
  const QC q = make_qc(pd->q, pd->mu64);
  const TW* twf = tw_arena + pd->tw_fwd_off;
  const TW* twi = tw_arena + pd->tw_inv_off;
  // (one workgroup = one prime: the branch is uniform; the sub-transform tables of a Proth-form prime hold 8-byte
  // entries, engine.hip conv_tables_sub)
#ifndef HX_NO_PROTH
  if (pd->proth) {
    conv_body<LOGN, ArProth>(lds, io, reinterpret_cast<const TWM*>(twf), reinterpret_cast<const TWM*>(twi), q);
    return;
  }
#endif
  conv_body<LOGN, ArShoup>(lds, io, twf, twi, q);
}

template <int LOGN, uint32_t SRC, uint32_t DST>
static hipError_t launch_conv(const ConvRowArgs& A, const ConvRows& R, int nunits, const PrimeDev* cprimes,
                              const TW* tw_arena, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  bool attr_set = false;   // (hxp::dyn_lds is idempotent per device)
  if (!attr_set) {
    hipError_t e = hxp::dyn_lds((const void*)ntt_conv_kernel<LOGN, SRC, DST>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess)
      return e;
    attr_set = true;
  }
  HX_LAUNCH((ntt_conv_kernel<LOGN, SRC, DST>), dim3((unsigned)nunits * A.batch), dim3(Geo<LOGN>::T), lds_bytes, st, A, R,
            cprimes, tw_arena);
  return hipGetLastError();
}
template <int LOGN>
static hipError_t launch_conv_n(const ConvRowArgs& A, const ConvRows& R, int nunits, const PrimeDev* cprimes,
                                const TW* tw_arena, hipStream_t st)
{
  if (A.src_mode == CONV_SRC_BLUE_PRE && A.dst_mode == CONV_DST_SUB)
    return launch_conv<LOGN, CONV_SRC_BLUE_PRE, CONV_DST_SUB>(A, R, nunits, cprimes, tw_arena, st);
  if (A.src_mode == CONV_SRC_SCATTER && A.dst_mode == CONV_DST_SUB)
    return launch_conv<LOGN, CONV_SRC_SCATTER, CONV_DST_SUB>(A, R, nunits, cprimes, tw_arena, st);
  if (A.src_mode == CONV_SRC_REV && A.dst_mode == CONV_DST_SUB)
    return launch_conv<LOGN, CONV_SRC_REV, CONV_DST_SUB>(A, R, nunits, cprimes, tw_arena, st);
  if (A.src_mode == CONV_SRC_REV && A.dst_mode == CONV_DST_FINAL)
    return launch_conv<LOGN, CONV_SRC_REV, CONV_DST_FINAL>(A, R, nunits, cprimes, tw_arena, st);
  return hipErrorInvalidValue;
}
// nunits = rows * split units of 2^logn points, times A.batch batch elements
hipError_t launch_conv_rows(int logn, const ConvRowArgs& A, const ConvRows& R, int nunits, const PrimeDev* cprimes,
                            const TW* tw_arena, hipStream_t st)
{
  switch (logn) {
    case 13: return launch_conv_n<13>(A, R, nunits, cprimes, tw_arena, st);
    case 14: return launch_conv_n<14>(A, R, nunits, cprimes, tw_arena, st);
    case 15: return launch_conv_n<15>(A, R, nunits, cprimes, tw_arena, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace hx

// ntt_kernels.hip -- power-of-two negacyclic NTT kernels for gfx950.
//
// One workgroup transforms one DoubleCRT row (one prime, one batch element)
// held entirely on-chip: 32 coefficients per thread in VGPRs, two LDS
// transposes (32-bit halves, 33/32-padded so both sides are bank-conflict
// free), twiddles streamed from L2-resident per-prime tables laid out so that
// every wave-load is contiguous.  HBM traffic = the algorithmic 16*N bytes.
// See ntt_core.h for the math and the phase functions.
#include <cstdlib>
#include "dev_common.h"
#include "prof.h"

#include "ntt_kernel_util.h"

// One translation unit per ring size.  The row kernels are straight-line code of ~7000 instructions each, in
// 3 ring sizes x 2 arithmetics x a dozen IO functors: compiled as one unit this file takes nine minutes.  Built
// with -DHX_NTT_ONLY=13|14|15 a unit instantiates the kernels of that ring size only and exports its entry points
// with the suffix _L13 / _L14 / _L15; ntt_dispatch.hip picks by logn (helib_amd/build.py compiles the three in
// parallel).  Without the macro the file is the whole thing under the plain names, as before.
#ifdef HX_NTT_ONLY
#define HX_SFX2(n, s) n##_L##s
#define HX_SFX1(n, s) HX_SFX2(n, s)
#define HX_ENTRY(n) HX_SFX1(n, HX_NTT_ONLY)
#else
#define HX_ENTRY(n) n
#endif
#if !defined(HX_NTT_ONLY) || HX_NTT_ONLY == 13
#define HX_SZ_13(...) case 13: return __VA_ARGS__;
#else
#define HX_SZ_13(...)
#endif
#if !defined(HX_NTT_ONLY) || HX_NTT_ONLY == 14
#define HX_SZ_14(...) case 14: return __VA_ARGS__;
#else
#define HX_SZ_14(...)
#endif
#if !defined(HX_NTT_ONLY) || HX_NTT_ONLY == 15
#define HX_SZ_15(...) case 15: return __VA_ARGS__;
#else
#define HX_SZ_15(...)
#endif
#define HX_SZ(N, ...) HX_SZ_##N(__VA_ARGS__)

namespace hx {

// Row accessor through a buffer resource: the row base lives in 4 SGPRs, the
// thread part of the address is ONE VGPR (tid*8) and the per-element constant
// goes into the scalar offset, so the 32 loads / 32 stores of a thread need no
// 64-bit address arithmetic and no address registers.
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v2i32 __attribute__((ext_vector_type(2)));
__device__ v2i32 hx_buffer_load_v2(v4i32 rsrc, int voffset, int soffset,
                                   int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");
__device__ void hx_buffer_store_v2(v2i32 data, v4i32 rsrc, int voffset, int soffset,
                                   int aux) __asm("llvm.amdgcn.raw.buffer.store.v2i32");

// Cache policy of rows that are read once or written once (the polynomial rows themselves): non-temporal, so that
// what IS shared inside an XCD's L2 -- twiddle tables, the x / S words every kept row of an element re-reads, the
// operand rows the three product parts share -- is not pushed out by them.  (aux bit 1 = nt on gfx940+.)
#ifdef HX_NO_NT
#define HX_NT 0
#define HX_NT_ASM ""
#else
#define HX_NT 2
#define HX_NT_ASM " nt"
#endif

__device__ v4i32 hx_buffer_load_v4(v4i32 rsrc, int voffset, int soffset,
                                   int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");

__device__ __forceinline__ v4i32 make_rsrc(const uint64_t* row, unsigned bytes)
{
  uint64_t a = (uint64_t)row;
  v4i32 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  r.z = (int)bytes;
  r.w = 0x00020000;  // gfx9 family: raw buffer, dword elements
  return r;
}
// Twiddle table of one prime behind a buffer resource (see tw_uni / tw_vec in ntt_core.h):
// the per-lane part of the address is one 32-bit VGPR (lane*16), the table position of the
// (stage, group) goes into the scalar offset, so a fetch costs one v_mov (the launder copy) and
// no 64-bit address arithmetic.
struct BufTw {
  v4i32 r;
  const TW* p;
  __device__ explicit BufTw(const TW* tw) : r(make_rsrc(reinterpret_cast<const uint64_t*>(tw), 0xfffffff0u)), p(tw) {}
};
__device__ __forceinline__ TW tw_uni(const BufTw& t, unsigned i) { return t.p[i]; }
__device__ __forceinline__ TW tw_vec(const BufTw& t, unsigned base, unsigned lane, unsigned off, uint32_t dep)
{
  int vo = (int)(lane * 16u);
  asm volatile("" : "+v"(vo) : "v"(dep));
  const v4i32 x = hx_buffer_load_v4(t.r, vo, (int)((base + off) * 16u), 0);
  TW w;
  w.w = ((uint64_t)(uint32_t)x.y << 32) | (uint32_t)x.x;
  w.wp = ((uint64_t)(uint32_t)x.w << 32) | (uint32_t)x.z;
  return w;
}

// LB: bound (in units of q) of the words the row holds on entry -- 1: canonical residues; 8: the lazy output of
// the exact-RNS kernels (break_digits_fast_kernel<., true>: values in [0,6q) they did not finish reducing, because
// the forward transform that reads them takes any bound up to 12 and tracks it at compile time)
// Rows of Proth-form primes under LB = 8: their only producer is the digit kernel's Proth-form target sum
// (rns_kernels.h TgtRec::mont: mont_redc128, below q (n/16 + 1 + 2^-32) < 2q for digits of up to eight primes) -- the
// engine leaves digit words unreduced only when that form is on (engine.hip digits_lazy_ok) -- so the Proth-form passes
// start from bound 2 and take one conditional-subtraction stage less (three instead of four at N = 2^14 / 2^15).
template <int LB>
struct BufIOT {
  static constexpr int LOAD_BOUND = LB;
#ifndef HX_DIGITS_LB8
  template <class AR>
  static constexpr int load_bound() { return (LB == 8 && AR::PROTH) ? 2 : LB; }
#endif
  static constexpr bool LAZY_STORE = false;
  static constexpr bool PIPELINED = false;
  struct StorePrefetch {};
  v4i32 rin, rout;
  __device__ BufIOT(const uint64_t* in_row, uint64_t* out_row, unsigned bytes)
      : rin(make_rsrc(in_row, bytes)), rout(make_rsrc(out_row, bytes))
  {
  }
  __device__ __forceinline__ uint64_t load(unsigned tid, unsigned c) const
  {
    v2i32 r = hx_buffer_load_v2(rin, (int)(tid * 8u), (int)(c * 8u), HX_NT);
    return ((uint64_t)(uint32_t)r.y << 32) | (uint32_t)r.x;
  }
  __device__ __forceinline__ void store(unsigned tid, unsigned c, uint64_t v) const
  {
    v2i32 d;
    d.x = (int)(uint32_t)v;
    d.y = (int)(uint32_t)(v >> 32);
    hx_buffer_store_v2(d, rout, (int)(tid * 8u), (int)(c * 8u), HX_NT);
  }
  __device__ __forceinline__ TW last_tw(TW def, int) const { return def; }
  __device__ __forceinline__ TWM last_tw(TWM def, int) const { return def; }
};

// inverse transform of the dropped row: stores x = F * iNTT(row) in [0,qd)  (F = 1 without the
// fused mod-up; otherwise F rides on the last stage's N^-1 twiddles, no extra multiplication).
// The delta preparation S(x) is a separate element-wise kernel (below): inside this store its
// ~250 instructions per element and dozen uniform 64-bit constants pushed the kernel into scratch.
struct InvPrepIO {
  static constexpr int LOAD_BOUND = 1;
  static constexpr bool LAZY_STORE = false;
  static constexpr bool PIPELINED = false;
  struct StorePrefetch {};
  v4i32 rin, rx;
  TW upS, upN;
  uint32_t has_up;
  __device__ InvPrepIO(const uint64_t* in_row, const ModDownPrep& p, size_t boff, unsigned bytes)
      : rin(make_rsrc(in_row, bytes)), rx(make_rsrc(p.xs + boff, bytes)), upS(p.upS), upN(p.upN),
        has_up(p.has_up)
  {
  }
  __device__ __forceinline__ uint64_t load(unsigned tid, unsigned c) const
  {
    v2i32 r = hx_buffer_load_v2(rin, (int)(tid * 8u), (int)(c * 8u), HX_NT);
    return ((uint64_t)(uint32_t)r.y << 32) | (uint32_t)r.x;
  }
  __device__ __forceinline__ void store(unsigned tid, unsigned c, uint64_t x) const
  {
    v2i32 d;
    d.x = (int)(uint32_t)x;
    d.y = (int)(uint32_t)(x >> 32);
    hx_buffer_store_v2(d, rx, (int)(tid * 8u), (int)(c * 8u), 0);
  }
  __device__ __forceinline__ TW last_tw(TW def, int which) const
  {
    return has_up ? (which ? upN : upS) : def;
  }
  __device__ __forceinline__ TWM last_tw(TWM def, int which) const   // (a Proth-form dropped prime: .wp holds w 2^64 mod qd)
  {
    return has_up ? (which ? upN.wp : upS.wp) : def;
  }
};

// delta = x - qd*S:  S = [x > (qd-1)/2]  (centring, src/DoubleCRT.cpp:1098-1099)
//                      + balanced((delta0 mod p) * qd^-1 mod p)  (ptxtSpace correction, :1485-1508)
__device__ __forceinline__ int64_t moddown_S_of(const ModDownPrep& P, uint64_t x)
{
  const bool neg = x > P.half;
  int64_t S = neg ? 1 : 0;
  if (P.ptxt > 1) {
    const uint64_t p = P.ptxt;
    uint64_t r = red64(x, p, P.ptxt_mu64);
    if (neg)
      r = sub_mod(r, P.qd_mod_p, p);  // delta mod p, non-negative
    if (r != 0) {
      // (uniform: a plaintext space below 2^32 -- the product fits a word and red64 replaces the 128-bit Barrett)
      const uint64_t dm = (p >> 32) == 0 ? red64(r * P.qdinv_mod_p, p, P.ptxt_mu64)
                                         : mul_mod(r, P.qdinv_mod_p, p, P.ptxt_mu, P.ptxt_k);
      const uint64_t p2 = p >> 1;
      const bool sub_p = dm > p2 || (((p & 1) == 0) && dm == p2 && neg);
      S += sub_p ? (int64_t)dm - (int64_t)p : (int64_t)dm;
    }
  }
  return S;
}
// n2 = pairs of words (a block of whole rows: even, 16-byte aligned)
// (a template only so that the per-ring-size translation units may each hold a copy: kernels of templates are
// merged by the linker, a plain __global__ function defined three times is a duplicate symbol)
template <int UNUSED = 0>
__global__ void __launch_bounds__(256)
moddown_S_kernel(ModDownPrep P, size_t n2)
{
  const ulonglong2* xs = reinterpret_cast<const ulonglong2*>(P.xs);
  longlong2* S = reinterpret_cast<longlong2*>(P.S);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const ulonglong2 x = xs[i];
    longlong2 s;
    s.x = moddown_S_of(P, x.x);
    s.y = moddown_S_of(P, x.y);
    S[i] = s;
  }
}

// forward transform of a kept row.  With inv = qd^-1 mod q_r and delta = x - qd*S:
//   load  = delta * inv = x*inv - S  (mod q_r)      (qd*inv = 1: S needs no multiplication,
//                                                    and x < 2^64 needs no reduction before Shoup)
//   store = c_r*cf - NTT(load)   with cf = inv (plain scale-down) or F*inv (mod-up folded in)
// PLAIN: the transform's input is a ready coefficient row (delta * P^-1 mod q_r from the
// basis-extension kernel, several dropped primes) instead of x*inv - S
template <bool PLAIN>
struct ModDownIO {
  // the load is x*inv (shoup4: [0,4q)) plus q - S in (0,2q); the store takes the lazy
  // transform output and normalises once, after the subtraction
  static constexpr int LOAD_BOUND = PLAIN ? 8 : 6;   // (PLAIN: ExtArgs::lazy_out words, [0,8q))
  // Proth-form rows: x*inv is a Montgomery product below 2q (mont_acc), plus q - S in (0,2q)
  // PLAIN on a Proth-form row: the extension word is a mont_redc128 result below 3q plus the plaintext-space correction
  // (rns_extend_fast_one, TgtRec::mont) -- below 4q; the engine asks for unreduced words only when that form is on
  template <class AR>
#ifndef HX_DIGITS_LB8
  static constexpr int load_bound() { return AR::PROTH ? 4 : (PLAIN ? 8 : 6); }
#else
  static constexpr int load_bound() { return PLAIN ? 8 : (AR::PROTH ? 4 : 6); }
#endif
  static constexpr bool LAZY_STORE = true;
  // Element IO is software-pipelined in groups of IOG elements.  Round 1 loaded x, S (and, in the
  // store, c_r) inside the same scheduling region as the ~40 instructions that consume them, one
  // group of four at a time: 16 exposed memory round trips per workgroup, 139 ns per row against
  // 89 ns for the plain transform with only 1.3x its arithmetic.  Now all 32 x words are requested
  // first (they land in the coefficient file itself), S runs one group ahead of the arithmetic
  // in a double buffer, and c_r is requested one group ahead starting BEFORE the last register
  // pass (StorePrefetch).
  static constexpr bool PIPELINED = !PLAIN;
  static constexpr int IOG = 4;
  struct StorePrefetch {
    uint64_t c[IOG];
  };
  v4i32 rx, rS, rc, ro;
  TW inv, cf;
  uint64_t q;
  // x_row: the x block of this (poly, batch) element, or -- PLAIN -- its delta row for this prime
  __device__ ModDownIO(const uint64_t* x_row, const int64_t* S_row, const ModDownRow& R, const uint64_t* c_row,
                       uint64_t* o_row, unsigned bytes, uint64_t q_)
      : rx(make_rsrc(x_row, bytes)), rS(make_rsrc((const uint64_t*)S_row, bytes)),
        rc(make_rsrc(c_row, bytes)), ro(make_rsrc(o_row, bytes)), inv(R.inv), cf(R.cf), q(q_)
  {
  }
  __device__ __forceinline__ uint64_t load(unsigned tid, unsigned c) const   // (PLAIN: the delta row, read once)
  {
    v2i32 a = hx_buffer_load_v2(rx, (int)(tid * 8u), (int)(c * 8u), HX_NT);
    return ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x;
  }
  static __device__ __forceinline__ uint64_t ld(const v4i32& r, unsigned tid, unsigned c)
  {
    v2i32 a = hx_buffer_load_v2(r, (int)(tid * 8u), (int)(c * 8u), 0);
    return ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x;
  }
  template <int LOGN, class AR>
  __device__ __forceinline__ void load_all(unsigned tid, uint64_t (&v)[32], const QC& qc) const
  {
    static_for<0, 32>([&](auto E) {
      constexpr int e = decltype(E)::value;
      v[e] = ld(rx, tid, coef_const<LOGN>(e));
    });
    uint64_t sb[2][IOG];
    static_for<0, IOG>([&](auto J) {
      constexpr int j = decltype(J)::value;
      sb[0][j] = ld(rS, tid, coef_const<LOGN>(j));
    });
    HX_SCHED_FENCE();
    static_for<0, 32 / IOG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      if constexpr (g + 1 < 32 / IOG)
        static_for<0, IOG>([&](auto J) {
          constexpr int j = decltype(J)::value;
          sb[(g + 1) & 1][j] = ld(rS, tid, coef_const<LOGN>((g + 1) * IOG + j));
        });
      static_for<0, IOG>([&](auto J) {
        constexpr int j = decltype(J)::value, e = g * IOG + j;
        // x*inv in [0,4q) (Proth-form rows: (0,2q)) plus q - S in (0,2q): |S| <= ptxtSpace/2 + 1 < q (host-checked),
        // so the two's-complement difference q - S is the right positive number whatever the sign of S,
        // and it rides on the multiply-add chain of the product as its addend -- no sign handling
        if constexpr (AR::PROTH)
          v[e] = mont_acc(v[e], inv.wp, qc, q - sb[g & 1][j]);   // (x < qd < 2^60: below the operand limit)
        else
          v[e] = shoup4_acc(v[e], inv, qc.nq, q - sb[g & 1][j]);
      });
      HX_SCHED_FENCE();
    });
  }
  // A load the compiler can neither sink into the branch that consumes it nor delay to the end of
  // its scheduling region (it did both with the intrinsic: the c_r requests ended up after the
  // last register pass and after the previous group's stores).  The instruction is issued where
  // it is written; the value exists after the matching pinned_wait<N> (N = memory operations
  // issued after it that may still be in flight -- the vector memory counter retires in order).
  static __device__ __forceinline__ uint64_t ld_pinned(const v4i32& r, unsigned tid, unsigned c)
  {
    uint64_t x;
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" HX_NT_ASM
                 : "=v"(x)
                 : "v"((int)(tid * 8u)), "s"(r), "s"((int)(c * 8u))
                 : "memory");
    return x;
  }
  template <int N>
  static __device__ __forceinline__ void pinned_wait(uint64_t (&x)[IOG])
  {
    static_assert(IOG == 4, "operand list");
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "n"(N) : "memory");
  }
  template <int LOGN>
  __device__ __forceinline__ void store_prefetch(unsigned tid, StorePrefetch& pre) const
  {
    static_for<0, IOG>([&](auto J) {
      constexpr int j = decltype(J)::value;
      pre.c[j] = ld_pinned(rc, tid, eval_const<LOGN>(j));
    });
  }
  // the whole store loop of the forward transform (v[i] in [0, B q), evaluation order)
  template <int LOGN, class AR, int B, bool EST>
  __device__ __forceinline__ void store_all(unsigned tid, uint64_t (&v)[32], const QC& qc, StorePrefetch& pre) const
  {
    // Proth-form rows: c_r*cf as a Montgomery product in (0,2q) riding on B q - x > 0 -- no conditional
    // subtraction of x first (the Shoup form takes x below 8q so that 8q - x is positive), one normalisation
    static_assert(!AR::PROTH || B <= 14, "B q - x + 2q must stay below 16q");
    const uint64_t Bq = (uint64_t)B * qc.q;
    // (rows added by a fused mod-up have no c_r: the host gives them cf = 0, so whatever their
    // never-initialised slot holds is multiplied away and the output is -NTT(.); no branch here --
    // a branch would let the compiler sink the prefetched loads below the last register pass)
    uint64_t cb[2][IOG];
    static_for<0, IOG>([&](auto J) { cb[0][decltype(J)::value] = pre.c[decltype(J)::value]; });
    static_for<0, 32 / IOG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      constexpr bool more = g + 1 < 32 / IOG;
      if constexpr (more)
        static_for<0, IOG>([&](auto J) {
          constexpr int j = decltype(J)::value;
          cb[(g + 1) & 1][j] = ld_pinned(rc, tid, eval_const<LOGN>((g + 1) * IOG + j));
        });
      // in flight behind group g's words: the IOG stores of group g-1 and the IOG loads of group g+1
      pinned_wait<(g > 0 ? IOG : 0) + (more ? IOG : 0)>(cb[g & 1]);
      static_for<0, IOG>([&](auto J) {
        constexpr int j = decltype(J)::value, i = g * IOG + j;
        uint64_t x = v[i];
        if constexpr (AR::PROTH) {
          put(tid, eval_const<LOGN>(i), norm_from<B + 2, EST>(mont_acc(cb[g & 1][j], cf.wp, qc, Bq - x), qc));
        } else {
          if constexpr (B > 8)
            x = csub(x, qc.q8);                            // [0,8q)
          // c_r*cf - x as a value in (0,12q), normalised once
          put(tid, eval_const<LOGN>(i), norm_from<12, EST>(shoup4_acc(cb[g & 1][j], cf, qc.nq, qc.q8 - x), qc));
        }
      });
      HX_SCHED_FENCE();
    });
  }
  // (the store is written out as well: the s_waitcnt vmcnt(N) of pinned_wait counts the memory operations issued
  // after the load it waits for -- the pinned loads of the next group AND these stores -- so each of them must be
  // exactly one instruction, at the place it is written; a compiler-emitted store could be merged, split or moved)
  __device__ __forceinline__ void put(unsigned tid, unsigned c, uint64_t o) const
  {
    asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" HX_NT_ASM
                 :
                 : "v"(o), "v"((int)(tid * 8u)), "s"(ro), "s"((int)(c * 8u))
                 : "memory");
  }
  __device__ __forceinline__ TW last_tw(TW def, int) const { return def; }
  __device__ __forceinline__ TWM last_tw(TWM def, int) const { return def; }
};

// (tile shape and work map of the mod-down apply kernels: work_map.h -- md_tile, md_work)

template <int LOGN, bool INV, class AR, class IO>
__device__ __forceinline__ void ntt_body_ar(uint32_t* lds, const IO& io, const typename AR::Tw* tw, const QC& q)
{
  using R = RowNTT<LOGN, AR>;
  uint64_t v[32];
  uint32_t nl[32];
  const unsigned w = wave_index();
  if constexpr (!INV) {
    R::template fwd<0>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template fwd<1>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template fwd<2>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template fwd<3>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template fwd<4>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template fwd<5>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template fwd<6>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template fwd<7>(fresh_tid(w), v, nl, lds, io, tw, q);
  } else {
    R::template inv<0>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template inv<1>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template inv<2>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template inv<3>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template inv<4>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template inv<5>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template inv<6>(fresh_tid(w), v, nl, lds, io, tw, q);
    __syncthreads();
    R::template inv<7>(fresh_tid(w), v, nl, lds, io, tw, q);
  }
}
// One row, in the arithmetic of its prime: rows of Proth-form primes (q = 1 mod 2^32: every prime of the
// benchmark chains, PrimeDev::proth) run the word-wise Montgomery butterflies on 8-byte table entries, any other
// prime the Shoup butterflies on {w, w'} pairs.  The branch is uniform over the workgroup (one row, one prime)
// and taken once; a launch may mix the two kinds of row.
template <int LOGN, bool INV, class IO>
__device__ __forceinline__ void ntt_body(uint32_t* lds, const IO& io, const TW* tw_ptr, const PrimeDev* pd)
{
  const QC q = make_qc(pd->q, pd->mu64);  // (the phase functions' last argument)
#ifndef HX_NO_PROTH
  if (pd->proth) {
    ntt_body_ar<LOGN, INV, ArProth>(lds, io, reinterpret_cast<const TWM*>(tw_ptr), q);
    return;
  }
#endif
  ntt_body_ar<LOGN, INV, ArShoup>(lds, io, tw_ptr, q);
}

// scaleDownToSet, one dropped prime: inverse transform of its row with the delta preparation
template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::T, HX_NTT_MINWAVES(LOGN))
ntt_moddown_prep_kernel(PolyBases polys, int row, int prime, int batch, ModDownPrep P,
                        const PrimeDev* __restrict__ primes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int b = (int)(blockIdx.x % (unsigned)batch), pi = (int)(blockIdx.x / (unsigned)batch);
  const PrimeDev* pd = primes + prime;
  const size_t N = Geo<LOGN>::N;
  const uint64_t* in = poly_base(polys, (unsigned)pi);
  const size_t pstride = P.poly_stride ? (size_t)P.poly_stride : (size_t)batch * N;
  const InvPrepIO io(in + ((size_t)row * batch + b) * N, P, (size_t)pi * pstride + (size_t)b * N,
                     (unsigned)N * 8u);
  ntt_body<LOGN, true>(lds, io, tw_arena + pd->tw_inv_off, pd);
}
// the same for up to MD_MAXDROP dropped primes at once (the several-primes mod-switch): one launch of
// ndrop * polys * batch workgroups instead of ndrop launches of polys * batch -- a single such launch
// is one round of resident workgroups (512 at batch 128), i.e. latency- and not throughput-bound
template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::T, HX_NTT_MINWAVES(LOGN))
ntt_moddown_prep_multi_kernel(PolyBases polys, PrepMulti M, int batch, ModDownPrep P,
                              const PrimeDev* __restrict__ primes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const unsigned per = (unsigned)polys.n * (unsigned)batch;
  const unsigned j = blockIdx.x / per, rem = blockIdx.x % per;
  const int b = (int)(rem % (unsigned)batch), pi = (int)(rem / (unsigned)batch);
  const PrimeDev* pd = primes + uniform_u16(M.prime, j);
  const int row = (int)uniform_u16(M.row, j);
  const size_t N = Geo<LOGN>::N;
  const uint64_t* in = poly_base(polys, (unsigned)pi);
  ModDownPrep Pj = P;
  Pj.xs = P.xs + (size_t)j * (size_t)batch * N;
  Pj.upS = M.up[2 * j];
  Pj.upN = M.up[2 * j + 1];
  const InvPrepIO io(in + ((size_t)row * batch + b) * N, Pj, (size_t)pi * (size_t)P.poly_stride + (size_t)b * N,
                     (unsigned)N * 8u);
  ntt_body<LOGN, true>(lds, io, tw_arena + pd->tw_inv_off, pd);
}
// ... forward transform of delta on every kept row, subtract + divide in the store
template <int LOGN, bool PLAIN>
__global__ void __launch_bounds__(Geo<LOGN>::T, HX_NTT_MINWAVES(LOGN))
ntt_moddown_apply_kernel(PolyBases polys, PolyBases outs, NttRows rows, int nkeep, int batch, ModDownApply A,
                         const PrimeDev* __restrict__ primes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
#ifdef HX_MD_OLDMAP
  // (row, poly, batch) order so that a chunk of consecutive work items shares one prime
  const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
  const int b = (int)(wid % (unsigned)batch);
  const unsigned rp = wid / (unsigned)batch;
  const unsigned pi = rp % (unsigned)polys.n, ri = rp / (unsigned)polys.n;
#else
  // 2-D XCD-aware tiling (work_map.h: md_work)
  const MdWork Wk = md_work(blockIdx.x, (unsigned)nkeep, (unsigned)polys.n * (unsigned)batch);
  if (!Wk.active)
    return;  // padding workgroup of an uneven tile (whole workgroup, before any barrier)
  const unsigned ri = Wk.ri, pb = Wk.pb;
  const int b = (int)(pb % (unsigned)batch);
  const unsigned pi = pb / (unsigned)batch;
#endif
  const PrimeDev* pd = primes + uniform_u16(rows.prime, ri);
  const size_t N = Geo<LOGN>::N;
  const ModDownRow R = A.rows[ri];
  // c_r is read from the poly's input slab and the result written to its output slab (the same
  // slab unless the poly was a lazy copy: engine.hip, hx_poly::Share).  A row the fused mod-up adds
  // has no c_r (cf = 0): its never-initialised OUTPUT slot stands in, the input slab may be
  // too small to have that row at all.
  uint64_t* odata = poly_base(outs, pi);
  const uint64_t* idata = R.mode == 2 ? odata : poly_base(polys, pi);
  const size_t eoff = ((size_t)pi * batch + b) * N;
  const uint64_t* xrow = PLAIN ? A.delta + (size_t)pi * (size_t)A.delta_poly_stride + ((size_t)ri * batch + b) * N
                               : A.xs + eoff;
  const ModDownIO<PLAIN> io(xrow, PLAIN ? nullptr : A.S + eoff, R,
                            idata + ((size_t)(R.mode == 2 ? R.out_row : uniform_u16(rows.row, ri)) * batch + b) * N,
                            odata + ((size_t)R.out_row * batch + b) * N, (unsigned)N * 8u, pd->q);
  ntt_body<LOGN, false>(lds, io, tw_arena + pd->tw_fwd_off, pd);
}

// ---- tensorProduct folded into the single-prime mod-switch (dev_common.h: TensorSrc) ----
// S mod q for S < 8 q^2: the approximate-quotient Barrett of rns_kernels.h (red128_q8: 7 word multiplications, the
// quotient estimate at most 5 short), restated here for this translation unit
__device__ __forceinline__ uint64_t tensor_red128(u128 S, uint64_t q, uint64_t mu63, uint32_t k)
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
// product part `part` (0: a0 b0, 1: a0 b1 + a1 b0, 2: a1 b1) of one coefficient, canonical; part 1 takes the
// two 128-bit products in ONE reduction (x < 2 q^2).  mu = PrimeDev::mu63.
__device__ __forceinline__ uint64_t tensor_value(unsigned part, uint64_t a0, uint64_t a1, uint64_t b0, uint64_t b1,
                                                 uint64_t q, uint64_t mu, uint32_t k)
{
  if (part == 1)
    return tensor_red128((u128)a0 * b1 + (u128)a1 * b0, q, mu, k);
  return tensor_red128((u128)a0 * b0, q, mu, k);   // (the caller hands part 2 its operands in the a0 / b0 slots)
}
// inverse transform of the dropped row of a product part: the row is formed from the operands' rows on load
struct InvPrepTensorIO {
  static constexpr int LOAD_BOUND = 1;
  static constexpr bool LAZY_STORE = false;
  static constexpr bool PIPELINED = false;
  struct StorePrefetch {};
  v4i32 ra, rb, rc, rd, rx;   // ra/rb: the operand pair of parts 0 and 2 (or a0, b1 of part 1); rc/rd: a1, b0 (part 1)
  TW upS, upN;
  uint32_t has_up, part, k;
  uint64_t q, mu;
  __device__ InvPrepTensorIO(const TensorSrc& T, unsigned part_, size_t roff, const ModDownPrep& p, size_t xoff,
                             unsigned bytes, const PrimeDev* pd)
      : ra(make_rsrc((part_ == 2 ? T.a1 : T.a0) + roff, bytes)),
        rb(make_rsrc((part_ == 0 ? T.b0 : T.b1) + roff, bytes)),
        rc(make_rsrc(T.a1 + roff, bytes)), rd(make_rsrc(T.b0 + roff, bytes)),
        rx(make_rsrc(p.xs + xoff, bytes)), upS(p.upS), upN(p.upN), has_up(p.has_up), part(part_), k(pd->k), q(pd->q),
        mu(pd->mu63)
  {
  }
  static __device__ __forceinline__ uint64_t ld(const v4i32& r, unsigned tid, unsigned c)
  {
    v2i32 a = hx_buffer_load_v2(r, (int)(tid * 8u), (int)(c * 8u), 0);
    return ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x;
  }
  __device__ __forceinline__ uint64_t load(unsigned tid, unsigned c) const
  {
    const uint64_t x = ld(ra, tid, c), y = ld(rb, tid, c);
    if (part == 1)
      return tensor_value(1, x, ld(rc, tid, c), ld(rd, tid, c), y, q, mu, k);   // a0 b1 + a1 b0
    return tensor_value(0, x, 0, y, 0, q, mu, k);
  }
  __device__ __forceinline__ void store(unsigned tid, unsigned c, uint64_t x) const
  {
    v2i32 d;
    d.x = (int)(uint32_t)x;
    d.y = (int)(uint32_t)(x >> 32);
    hx_buffer_store_v2(d, rx, (int)(tid * 8u), (int)(c * 8u), 0);
  }
  __device__ __forceinline__ TW last_tw(TW def, int which) const { return has_up ? (which ? upN : upS) : def; }
  __device__ __forceinline__ TWM last_tw(TWM def, int which) const { return has_up ? (which ? upN.wp : upS.wp) : def; }
};
template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::T, HX_NTT_MINWAVES(LOGN))
ntt_moddown_prep_tensor_kernel(TensorSrc T, int row, int prime, int batch, ModDownPrep P,
                               const PrimeDev* __restrict__ primes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int b = (int)(blockIdx.x % (unsigned)batch), pi = (int)(blockIdx.x / (unsigned)batch);   // pi = product part
  const PrimeDev* pd = primes + prime;
  const size_t N = Geo<LOGN>::N;
  const InvPrepTensorIO io(T, (unsigned)pi, ((size_t)row * batch + b) * N, P, ((size_t)pi * batch + b) * N,
                           (unsigned)N * 8u, pd);
  ntt_body<LOGN, true>(lds, io, tw_arena + pd->tw_inv_off, pd);
}

// the same for up to MD_MAXDROP dropped primes at once (ntt_moddown_prep_multi_kernel with the product parts as input)
template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::T, HX_NTT_MINWAVES(LOGN))
ntt_moddown_prep_multi_tensor_kernel(TensorSrc T, PrepMulti M, int batch, ModDownPrep P,
                                     const PrimeDev* __restrict__ primes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const unsigned per = 3u * (unsigned)batch;
  const unsigned j = blockIdx.x / per, rem = blockIdx.x % per;
  const int b = (int)(rem % (unsigned)batch), pi = (int)(rem / (unsigned)batch);   // pi = product part
  const PrimeDev* pd = primes + uniform_u16(M.prime, j);
  const int row = (int)uniform_u16(M.row, j);
  const size_t N = Geo<LOGN>::N;
  ModDownPrep Pj = P;
  Pj.xs = P.xs + (size_t)j * (size_t)batch * N;
  Pj.upS = M.up[2 * j];
  Pj.upN = M.up[2 * j + 1];
  const InvPrepTensorIO io(T, (unsigned)pi, ((size_t)row * batch + b) * N, Pj,
                           (size_t)pi * (size_t)P.poly_stride + (size_t)b * N, (unsigned)N * 8u, pd);
  ntt_body<LOGN, true>(lds, io, tw_arena + pd->tw_inv_off, pd);
}

// forward transform of a kept row whose c_r is a product part formed from the operands' rows in the store
// PLAIN (several dropped primes): the transform's input is the ready delta * P^-1 row of the basis extension
template <bool PLAIN>
struct ModDownTensorIO {
  static constexpr int LOAD_BOUND = PLAIN ? 8 : 6;   // (PLAIN: ExtArgs::lazy_out words, [0,8q))
  template <class AR>
#ifndef HX_DIGITS_LB8
  static constexpr int load_bound() { return AR::PROTH ? 4 : (PLAIN ? 8 : 6); }   // (PLAIN Proth rows: as ModDownIO)
#else
  static constexpr int load_bound() { return PLAIN ? 8 : (AR::PROTH ? 4 : 6); }
#endif
  static constexpr bool LAZY_STORE = true;
  static constexpr bool PIPELINED = !PLAIN;
  static constexpr int IOG = 4;
  struct StorePrefetch {};
  v4i32 rx, rS, ra, rb, rc, rd, ro;
  TW inv, cf;
  uint64_t cf_r2;   // Proth rows: cf 2^128 mod q (ModDownRow::cf_r2)
  uint64_t q, mu;
  uint32_t part, k;
  // operand rows: [row][batch][N] slabs of the four operands at `roff`; a row the fused mod-up adds has no operand
  // row (cf = 0): its own output row stands in for all four, whatever it holds is multiplied away
  __device__ ModDownTensorIO(const uint64_t* x_row, const int64_t* S_row, const ModDownRow& R, const TensorSrc& T,
                             unsigned part_, size_t roff, bool has_row, uint64_t* o_row, unsigned bytes, const PrimeDev* pd)
      : rx(make_rsrc(x_row, bytes)), rS(make_rsrc((const uint64_t*)S_row, bytes)),
        ra(make_rsrc(has_row ? (part_ == 2 ? T.a1 : T.a0) + roff : o_row, bytes)),
        rb(make_rsrc(has_row ? (part_ == 0 ? T.b0 : T.b1) + roff : o_row, bytes)),
        rc(make_rsrc(has_row ? T.a1 + roff : o_row, bytes)), rd(make_rsrc(has_row ? T.b0 + roff : o_row, bytes)),
        ro(make_rsrc(o_row, bytes)), inv(R.inv), cf(R.cf), cf_r2(R.cf_r2), q(pd->q), mu(pd->mu63), part(part_), k(pd->k)
  {
  }
  static __device__ __forceinline__ uint64_t ld(const v4i32& r, unsigned tid, unsigned c)
  {
    v2i32 a = hx_buffer_load_v2(r, (int)(tid * 8u), (int)(c * 8u), 0);
    return ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x;
  }
  __device__ __forceinline__ uint64_t load(unsigned tid, unsigned c) const   // (PLAIN: the delta row, read once)
  {
    v2i32 a = hx_buffer_load_v2(rx, (int)(tid * 8u), (int)(c * 8u), HX_NT);
    return ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x;
  }
  // the load of ModDownIO<false>: x*inv - S with S one group ahead
  template <int LOGN, class AR>
  __device__ __forceinline__ void load_all(unsigned tid, uint64_t (&v)[32], const QC& qc) const
  {
    static_for<0, 32>([&](auto E) {
      constexpr int e = decltype(E)::value;
      v[e] = ld(rx, tid, coef_const<LOGN>(e));
    });
    uint64_t sb[2][IOG];
    static_for<0, IOG>([&](auto J) {
      constexpr int j = decltype(J)::value;
      sb[0][j] = ld(rS, tid, coef_const<LOGN>(j));
    });
    HX_SCHED_FENCE();
    static_for<0, 32 / IOG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      if constexpr (g + 1 < 32 / IOG)
        static_for<0, IOG>([&](auto J) {
          constexpr int j = decltype(J)::value;
          sb[(g + 1) & 1][j] = ld(rS, tid, coef_const<LOGN>((g + 1) * IOG + j));
        });
      static_for<0, IOG>([&](auto J) {
        constexpr int j = decltype(J)::value, e = g * IOG + j;
        if constexpr (AR::PROTH)
          v[e] = mont_acc(v[e], inv.wp, qc, q - sb[g & 1][j]);
        else
          v[e] = shoup4_acc(v[e], inv, qc.nq, q - sb[g & 1][j]);
      });
      HX_SCHED_FENCE();
    });
  }
  // Operand words are requested one group (eight words: two coefficients of part 1, four of parts 0 / 2) AHEAD of
  // the arithmetic that consumes them, in a double buffer, with loads the compiler can neither sink nor delay
  // (ModDownIO::ld_pinned) and explicit vmcnt waits -- round 3 loaded a group and then waited for it, sixteen
  // exposed round trips per workgroup, which is what kept this kernel 19 % above its instruction floor
  // (profiles/r04_valu_floor_by_class.json).
  static __device__ __forceinline__ uint64_t ld_pinned(const v4i32& r, unsigned tid, unsigned c)
  {
    uint64_t x;
    asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(x) : "v"((int)(tid * 8u)), "s"(r), "s"((int)(c * 8u)) : "memory");
    return x;
  }
  template <int N>
  static __device__ __forceinline__ void pinned_wait(uint64_t (&x)[8])
  {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                 : "n"(N)
                 : "memory");
  }
  // the eight words of group g: part 1: (a0, b1, a1, b0) of coefficients 2g, 2g+1; else (x, y) of 4g .. 4g+3
  template <int LOGN, bool P1, int g>
  __device__ __forceinline__ void request(unsigned tid, uint64_t (&w)[8]) const
  {
    if constexpr (P1) {
      static_for<0, 2>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int i = 2 * g + j;
        const unsigned c = eval_const<LOGN>(i);
        w[4 * j + 0] = ld_pinned(ra, tid, c);
        w[4 * j + 1] = ld_pinned(rb, tid, c);
        w[4 * j + 2] = ld_pinned(rc, tid, c);
        w[4 * j + 3] = ld_pinned(rd, tid, c);
      });
    } else {
      static_for<0, 4>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int i = 4 * g + j;
        const unsigned c = eval_const<LOGN>(i);
        w[2 * j + 0] = ld_pinned(ra, tid, c);
        w[2 * j + 1] = ld_pinned(rb, tid, c);
      });
    }
  }
  template <int LOGN>
  __device__ __forceinline__ void store_prefetch(unsigned, StorePrefetch&) const {}
  template <int LOGN, class AR, int B, bool EST, bool P1>
  __device__ __forceinline__ void store_part(unsigned tid, uint64_t (&v)[32], const QC& qc) const
  {
    constexpr int CPG = P1 ? 2 : 4, NG = 32 / CPG;   // coefficients per group, groups
    static_assert(!AR::PROTH || B <= 14, "B q - x + 2q must stay below 16q");
    const uint64_t Bq = (uint64_t)B * qc.q;
    auto one = [&](auto I, uint64_t x0, uint64_t y0, uint64_t x1, uint64_t y1) {
      constexpr int i = decltype(I)::value;
      uint64_t x = v[i];
      if constexpr (AR::PROTH) {
        // the 128-bit product (part 1: the sum of two, below 2 q^2) goes through the Proth-form reduction alone --
        // two multiply-adds instead of the Barrett's seven multiplications and three conditional subtractions --
        // and carries 2^-64; the constant it is multiplied by next is cf 2^128 (ModDownRow::cf_r2), so that the
        // Montgomery product gives (a b) cf exactly; B q - x > 0 rides on it, one normalisation at the end
        const u128 S = P1 ? (u128)x0 * y0 + (u128)x1 * y1 : (u128)x0 * y0;   // (x0, y0, x1, y1) = (a0, b1, a1, b0)
        const uint64_t c2 = mont_redc128((uint64_t)S, (uint64_t)(S >> 64), qc);   // (0, 2q)
        put(tid, eval_const<LOGN>(i), norm_from<B + 2, EST>(mont_acc(c2, cf_r2, qc, Bq - x), qc));
      } else {
        const uint64_t c = P1 ? tensor_value(1, x0, x1, y1, y0, q, mu, k) : tensor_value(0, x0, 0, y0, 0, q, mu, k);
        if constexpr (B > 8)
          x = csub(x, qc.q8);
        put(tid, eval_const<LOGN>(i), norm_from<12, EST>(shoup4_acc(c, cf, qc.nq, qc.q8 - x), qc));
      }
    };
    uint64_t wb[2][8];
    request<LOGN, P1, 0>(tid, wb[0]);
    static_for<0, NG>([&](auto GI) {
      constexpr int g = decltype(GI)::value;
      constexpr bool more = g + 1 < NG;
      if constexpr (more)
        request<LOGN, P1, g + 1>(tid, wb[(g + 1) & 1]);
      // in flight behind group g's words: the CPG stores of group g-1 and the eight loads of group g+1
      pinned_wait<(g > 0 ? CPG : 0) + (more ? 8 : 0)>(wb[g & 1]);
      static_for<0, CPG>([&](auto J) {
        constexpr int j = decltype(J)::value;
        if constexpr (P1)
          one(std::integral_constant<int, g * CPG + j>{}, wb[g & 1][4 * j], wb[g & 1][4 * j + 1], wb[g & 1][4 * j + 2],
              wb[g & 1][4 * j + 3]);
        else
          one(std::integral_constant<int, g * CPG + j>{}, wb[g & 1][2 * j], wb[g & 1][2 * j + 1], 0, 0);
      });
      HX_SCHED_FENCE();
    });
  }
  template <int LOGN, class AR, int B, bool EST>
  __device__ __forceinline__ void store_all(unsigned tid, uint64_t (&v)[32], const QC& qc, StorePrefetch&) const
  {
    if (part == 1)   // (wave-uniform: one of the two loops runs)
      store_part<LOGN, AR, B, EST, true>(tid, v, qc);
    else
      store_part<LOGN, AR, B, EST, false>(tid, v, qc);
  }
  // (the store is written out as well: the s_waitcnt vmcnt(N) of pinned_wait counts the memory operations issued
  // after the load it waits for -- the pinned loads of the next group AND these stores -- so each of them must be
  // exactly one instruction, at the place it is written; a compiler-emitted store could be merged, split or moved)
  __device__ __forceinline__ void put(unsigned tid, unsigned c, uint64_t o) const
  {
    asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen" HX_NT_ASM
                 :
                 : "v"(o), "v"((int)(tid * 8u)), "s"(ro), "s"((int)(c * 8u))
                 : "memory");
  }
  __device__ __forceinline__ TW last_tw(TW def, int) const { return def; }
  __device__ __forceinline__ TWM last_tw(TWM def, int) const { return def; }
};
template <int LOGN, bool PLAIN>
__global__ void __launch_bounds__(Geo<LOGN>::T, HX_NTT_MINWAVES(LOGN))
ntt_moddown_apply_tensor_kernel(TensorSrc T, PolyBases outs, NttRows rows, int nkeep, int batch, ModDownApply A,
                                const PrimeDev* __restrict__ primes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const MdWork Wk = md_work(blockIdx.x, (unsigned)nkeep, 3u * (unsigned)batch);
  if (!Wk.active)
    return;
  const unsigned ri = Wk.ri, pb = Wk.pb;
#ifdef HX_TENSOR_PARTS_APART
  const int b = (int)(pb % (unsigned)batch);
  const unsigned pi = pb / (unsigned)batch;   // product part 0, 1, 2
#else
  // the three product parts of one batch element are neighbours in the XCD's dispatch order: an operand row is read
  // by two of them (a0: parts 0, 1; a1: 1, 2; ...), 3 * nr slots apart at most, i.e. while it is still in that XCD's L2
  const int b = (int)(pb / 3u);
  const unsigned pi = pb % 3u;                // product part 0, 1, 2
#endif
  const PrimeDev* pd = primes + uniform_u16(rows.prime, ri);
  const size_t N = Geo<LOGN>::N;
  const ModDownRow R = A.rows[ri];
  uint64_t* odata = poly_base(outs, pi);
  const size_t eoff = ((size_t)pi * batch + b) * N;
  const uint64_t* xrow = PLAIN ? A.delta + (size_t)pi * (size_t)A.delta_poly_stride + ((size_t)ri * batch + b) * N
                               : A.xs + eoff;
  const ModDownTensorIO<PLAIN> io(xrow, PLAIN ? nullptr : A.S + eoff, R, T, pi,
                                  ((size_t)uniform_u16(rows.row, ri) * batch + b) * N, R.mode != 2,
                                  odata + ((size_t)R.out_row * batch + b) * N, (unsigned)N * 8u, pd);
  ntt_body<LOGN, false>(lds, io, tw_arena + pd->tw_fwd_off, pd);
}

template <int LOGN, bool INV, int LB = 1>
__global__ void __launch_bounds__(Geo<LOGN>::T, HX_NTT_MINWAVES(LOGN))
ntt_row_kernel(const uint64_t* in, uint64_t* out, NttRows rows, int batch,
               const PrimeDev* __restrict__ primes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned ri = wid / (unsigned)batch;
  const int b = (int)(wid % (unsigned)batch);
  const int row = (int)uniform_u16(rows.row, ri);
  const PrimeDev* pd = primes + uniform_u16(rows.prime, ri);
  const size_t roff = ((size_t)row * batch + b) * (size_t)Geo<LOGN>::N;
  const TW* tw = tw_arena + (INV ? pd->tw_inv_off : pd->tw_fwd_off);

#ifdef HX_NTT_PTRIO
  const PtrIO io{in + roff, out + roff};
#else
  const BufIOT<LB> io(in + roff, out + roff, (unsigned)Geo<LOGN>::N * 8u);
#endif
  ntt_body<LOGN, INV>(lds, io, tw, pd);
}

// inverse transform of a(X) * b(X) given in evaluation form: the s^2 part of a tensor product (a1 b1) goes
// straight from the operands' rows to its coefficient rows (hx_mul_relin: toPoly side of breakIntoDigits)
// PROTH (rows of Proth-form primes, ntt_core.h ArProth): the 128-bit product goes through mont_redc128 alone -- two
// multiply-adds and two carries for the Barrett's seven multiplications and three conditional subtractions -- into
// (0, q (1/16 + 1 + 2^-32)): the inverse transform takes it as it is (bound 2, like the convolution kernel's pointwise
// product), and the 2^-64 it carries is given back by the last stage's constants (N^-1 twiddles x 2^128 by PrimeDev::r2)
template <bool PROTH>
struct MulLoadIO {
  static constexpr int LOAD_BOUND = 1;
  static constexpr bool LAZY_STORE = false;
  static constexpr bool PIPELINED = false;
  template <class AR>
  static constexpr int inv_load_bound() { return PROTH ? 2 : 1; }
  struct StorePrefetch {};
  v4i32 ra, rb, ro;
  uint64_t q, mu63, r2;
  uint32_t k;
  __device__ MulLoadIO(const uint64_t* a_row, const uint64_t* b_row, uint64_t* o_row, unsigned bytes, const PrimeDev* pd)
      : ra(make_rsrc(a_row, bytes)), rb(make_rsrc(b_row, bytes)), ro(make_rsrc(o_row, bytes)), q(pd->q), mu63(pd->mu63),
        r2(pd->r2), k(pd->k)
  {
  }
  __device__ __forceinline__ uint64_t load(unsigned tid, unsigned c) const
  {
    const v2i32 x = hx_buffer_load_v2(ra, (int)(tid * 8u), (int)(c * 8u), HX_NT);
    const v2i32 y = hx_buffer_load_v2(rb, (int)(tid * 8u), (int)(c * 8u), HX_NT);
    const uint64_t a = ((uint64_t)(uint32_t)x.y << 32) | (uint32_t)x.x, b = ((uint64_t)(uint32_t)y.y << 32) | (uint32_t)y.x;
    const u128 p = (u128)a * b;
    if constexpr (PROTH)
      return mont_redc128((uint64_t)p, (uint64_t)(p >> 64), make_qc(q, 0));
    else
      return tensor_red128(p, q, mu63, k);
  }
  __device__ __forceinline__ void store(unsigned tid, unsigned c, uint64_t v) const
  {
    v2i32 d;
    d.x = (int)(uint32_t)v;
    d.y = (int)(uint32_t)(v >> 32);
    hx_buffer_store_v2(d, ro, (int)(tid * 8u), (int)(c * 8u), HX_NT);
  }
  __device__ __forceinline__ TW last_tw(TW def, int) const { return def; }
  __device__ __forceinline__ TWM last_tw(TWM def, int) const
  {
    if constexpr (PROTH) {
      const QC qc = make_qc(q, 0);   // (q, qh, c1 are all this needs)
      return csub(mont_mul(def, r2, qc), q);
    } else {
      return def;
    }
  }
};
template <int LOGN>
__global__ void __launch_bounds__(Geo<LOGN>::T, HX_NTT_MINWAVES(LOGN))
ntt_inv_mul_kernel(const uint64_t* a, const uint64_t* b, uint64_t* out, NttRows rows, int batch,
                   const PrimeDev* __restrict__ primes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned ri = wid / (unsigned)batch;
  const int bb = (int)(wid % (unsigned)batch);
  const int row = (int)uniform_u16(rows.row, ri);
  const PrimeDev* pd = primes + uniform_u16(rows.prime, ri);
  const size_t N = Geo<LOGN>::N;
  const size_t roff = ((size_t)row * batch + bb) * N, ooff = ((size_t)ri * batch + bb) * N;   // output rows are compact
  const QC qc = make_qc(pd->q, pd->mu64);
  const TW* tw = tw_arena + pd->tw_inv_off;
#ifndef HX_NO_PROTH
  if (pd->proth) {   // (uniform: one workgroup, one prime)
#ifndef HX_INVMUL_BARRETT
    const MulLoadIO<true> io(a + roff, b + roff, out + ooff, (unsigned)N * 8u, pd);
#else
    const MulLoadIO<false> io(a + roff, b + roff, out + ooff, (unsigned)N * 8u, pd);   // (the A/B control: Barrett on load)
#endif
    ntt_body_ar<LOGN, true, ArProth>(lds, io, reinterpret_cast<const TWM*>(tw), qc);
    return;
  }
#endif
  const MulLoadIO<false> io(a + roff, b + roff, out + ooff, (unsigned)N * 8u, pd);
  ntt_body_ar<LOGN, true, ArShoup>(lds, io, tw, qc);
}
template <int LOGN>
static hipError_t launch_inv_mul(const uint64_t* a, const uint64_t* b, uint64_t* out, const NttRows& rows, int nrows,
                                 int batch, const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  bool attr_set = false;   // (hxp::dyn_lds is idempotent per device)
  if (!attr_set) {
    hipError_t e = hxp::dyn_lds((const void*)ntt_inv_mul_kernel<LOGN>, (int)lds_bytes);
    if (e != hipSuccess)
      return e;
    attr_set = true;
  }
  HX_LAUNCH((ntt_inv_mul_kernel<LOGN>), dim3((unsigned)nrows * (unsigned)batch), dim3(Geo<LOGN>::T), lds_bytes, st, a, b,
            out, rows, batch, primes, tw_arena);
  return hipGetLastError();
}
// out[i] (compact rows i < nrows) = inverse transform of a[rows.row[i]] * b[rows.row[i]]
hipError_t HX_ENTRY(launch_ntt_inv_mul_pow2)(int logn, const uint64_t* a, const uint64_t* b, uint64_t* out, const NttRows& rows,
                                   int nrows, int batch, const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
  switch (logn) {
    HX_SZ(13, launch_inv_mul<13>(a, b, out, rows, nrows, batch, primes, tw_arena, st))
    HX_SZ(14, launch_inv_mul<14>(a, b, out, rows, nrows, batch, primes, tw_arena, st))
    HX_SZ(15, launch_inv_mul<15>(a, b, out, rows, nrows, batch, primes, tw_arena, st))
  }
  return hipErrorInvalidValue;
}

template <int LOGN, bool INV, int LB = 1>
static hipError_t launch_one(const uint64_t* in, uint64_t* out, const NttRows& rows, int nrows,
                             int batch, const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
#ifdef HX_ROW_ONE_WG
  // (A/B probe: more LDS than half a CU's, i.e. ONE workgroup per CU -- what a row kernel with a 256-register budget
  // would get; profiles/r06_ab_row_kernel_one_workgroup_per_cu.json)
  static const size_t lds_bytes = LOGN == 14 ? (size_t)90 * 1024 : (size_t)Geo<LOGN>::LDS_WORDS * 4;
#else
  static const size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
#endif
  bool attr_set = false;   // (hxp::dyn_lds is idempotent per device)
  if (!attr_set) {
    hipError_t e = hxp::dyn_lds((const void*)ntt_row_kernel<LOGN, INV, LB>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess)
      return e;
    attr_set = true;
  }
  dim3 grid((unsigned)nrows * (unsigned)batch), block(Geo<LOGN>::T);
  HX_LAUNCH((ntt_row_kernel<LOGN, INV, LB>), grid, block, lds_bytes, st, in, out, rows, batch,
                     primes, tw_arena);
  return hipGetLastError();
}

template <int LOGN>
static hipError_t moddown_attrs()
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  bool attr_set = false;   // (hxp::dyn_lds is idempotent per device)
  if (!attr_set) {
    hipError_t e = hxp::dyn_lds((const void*)ntt_moddown_prep_kernel<LOGN>, (int)lds_bytes);
    if (e == hipSuccess)
      e = hxp::dyn_lds((const void*)ntt_moddown_prep_multi_kernel<LOGN>, (int)lds_bytes);
    if (e == hipSuccess)
      e = hxp::dyn_lds((const void*)ntt_moddown_apply_kernel<LOGN, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess)
      e = hxp::dyn_lds((const void*)ntt_moddown_apply_kernel<LOGN, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess)
      return e;
    attr_set = true;
  }
  return hipSuccess;
}
static unsigned moddown_apply_grid(unsigned npoly, unsigned nkeep, unsigned batch)
{
#ifdef HX_MD_OLDMAP
  return npoly * nkeep * batch;
#else
  return 8u * md_tile(nkeep, npoly * batch).per_xcd;
#endif
}

template <int LOGN>
static hipError_t launch_moddown(const PolyBases& polys, const PolyBases& outs, int drop_row, int drop_prime,
                                 const NttRows& keep, int nkeep, int batch, const ModDownPrep& P,
                                 const ModDownApply& A, const PrimeDev* primes, const TW* tw_arena,
                                 hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  hipError_t e = moddown_attrs<LOGN>();
  if (e != hipSuccess)
    return e;
  HX_LAUNCH((ntt_moddown_prep_kernel<LOGN>), dim3((unsigned)polys.n * (unsigned)batch),
                     dim3(Geo<LOGN>::T), lds_bytes, st, polys, drop_row, drop_prime, batch, P, primes,
                     tw_arena);
  {
    const size_t n = (size_t)polys.n * (size_t)batch * Geo<LOGN>::N;
    HX_LAUNCH((moddown_S_kernel<0>), dim3((unsigned)((n / 2 + 255) / 256 > 8192 ? 8192 : (n / 2 + 255) / 256)),
                       dim3(256), 0, st, P, n / 2);
  }
  HX_LAUNCH((ntt_moddown_apply_kernel<LOGN, false>),
                     dim3(moddown_apply_grid((unsigned)polys.n, (unsigned)nkeep, (unsigned)batch)),
                     dim3(Geo<LOGN>::T), lds_bytes, st, polys, outs, keep, nkeep, batch, A, primes, tw_arena);
  return hipGetLastError();
}

hipError_t HX_ENTRY(launch_moddown_pow2)(int logn, const PolyBases& data, const PolyBases& out, int drop_row,
                               int drop_prime, const NttRows& keep, int nkeep, int batch,
                               const ModDownPrep& P, const ModDownApply& A, const PrimeDev* primes,
                               const TW* tw_arena, hipStream_t st)
{
  switch (logn) {
    HX_SZ(13, launch_moddown<13>(data, out, drop_row, drop_prime, keep, nkeep, batch, P, A, primes, tw_arena, st))
    HX_SZ(14, launch_moddown<14>(data, out, drop_row, drop_prime, keep, nkeep, batch, P, A, primes, tw_arena, st))
    HX_SZ(15, launch_moddown<15>(data, out, drop_row, drop_prime, keep, nkeep, batch, P, A, primes, tw_arena, st))
  }
  return hipErrorInvalidValue;
}

// tensorProduct + single-prime mod-switch of the three product parts (TensorSrc): prep from the operands' dropped
// rows, S, apply forming c_r from the operands' kept rows
template <int LOGN>
static hipError_t moddown_tensor_attrs()
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  bool attr_set = false;   // (hxp::dyn_lds is idempotent per device)
  if (!attr_set) {
    hipError_t e = hxp::dyn_lds((const void*)ntt_moddown_prep_tensor_kernel<LOGN>, (int)lds_bytes);
    if (e == hipSuccess)
      e = hxp::dyn_lds((const void*)ntt_moddown_prep_multi_tensor_kernel<LOGN>, (int)lds_bytes);
    if (e == hipSuccess)
      e = hxp::dyn_lds((const void*)ntt_moddown_apply_tensor_kernel<LOGN, false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess)
      e = hxp::dyn_lds((const void*)ntt_moddown_apply_tensor_kernel<LOGN, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess)
      return e;
    attr_set = true;
  }
  return hipSuccess;
}
template <int LOGN>
static hipError_t launch_moddown_tensor(const TensorSrc& T, const PolyBases& outs, int drop_row, int drop_prime,
                                        const NttRows& keep, int nkeep, int batch, const ModDownPrep& P,
                                        const ModDownApply& A, const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  hipError_t ea = moddown_tensor_attrs<LOGN>();
  if (ea != hipSuccess)
    return ea;
  HX_LAUNCH((ntt_moddown_prep_tensor_kernel<LOGN>), dim3(3u * (unsigned)batch), dim3(Geo<LOGN>::T), lds_bytes, st, T,
            drop_row, drop_prime, batch, P, primes, tw_arena);
  {
    const size_t n = (size_t)3 * (size_t)batch * Geo<LOGN>::N;
    HX_LAUNCH((moddown_S_kernel<0>), dim3((unsigned)((n / 2 + 255) / 256 > 8192 ? 8192 : (n / 2 + 255) / 256)), dim3(256), 0, st, P,
              n / 2);
  }
  HX_LAUNCH((ntt_moddown_apply_tensor_kernel<LOGN, false>), dim3(moddown_apply_grid(3u, (unsigned)nkeep, (unsigned)batch)),
            dim3(Geo<LOGN>::T), lds_bytes, st, T, outs, keep, nkeep, batch, A, primes, tw_arena);
  return hipGetLastError();
}
hipError_t HX_ENTRY(launch_moddown_tensor_pow2)(int logn, const TensorSrc& T, const PolyBases& outs, int drop_row, int drop_prime,
                                      const NttRows& keep, int nkeep, int batch, const ModDownPrep& P,
                                      const ModDownApply& A, const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
  switch (logn) {
    HX_SZ(13, launch_moddown_tensor<13>(T, outs, drop_row, drop_prime, keep, nkeep, batch, P, A, primes, tw_arena, st))
    HX_SZ(14, launch_moddown_tensor<14>(T, outs, drop_row, drop_prime, keep, nkeep, batch, P, A, primes, tw_arena, st))
    HX_SZ(15, launch_moddown_tensor<15>(T, outs, drop_row, drop_prime, keep, nkeep, batch, P, A, primes, tw_arena, st))
  }
  return hipErrorInvalidValue;
}

// tensorProduct + several-primes mod-switch: the two launches below (prep_multi / apply_plain) with the product parts
// formed from the operands' rows (three parts, outs = their slabs)
template <int LOGN>
static hipError_t launch_prep_multi_tensor(const TensorSrc& T, const PrepMulti& M, int ndrop, int batch, const ModDownPrep& P,
                                           const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  hipError_t e = moddown_tensor_attrs<LOGN>();
  if (e != hipSuccess)
    return e;
  HX_LAUNCH((ntt_moddown_prep_multi_tensor_kernel<LOGN>), dim3((unsigned)ndrop * 3u * (unsigned)batch), dim3(Geo<LOGN>::T),
            lds_bytes, st, T, M, batch, P, primes, tw_arena);
  return hipGetLastError();
}
template <int LOGN>
static hipError_t launch_apply_plain_tensor(const TensorSrc& T, const PolyBases& outs, const NttRows& keep, int nkeep,
                                            int batch, const ModDownApply& A, const PrimeDev* primes, const TW* tw_arena,
                                            hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  hipError_t e = moddown_tensor_attrs<LOGN>();
  if (e != hipSuccess)
    return e;
  HX_LAUNCH((ntt_moddown_apply_tensor_kernel<LOGN, true>), dim3(moddown_apply_grid(3u, (unsigned)nkeep, (unsigned)batch)),
            dim3(Geo<LOGN>::T), lds_bytes, st, T, outs, keep, nkeep, batch, A, primes, tw_arena);
  return hipGetLastError();
}
hipError_t HX_ENTRY(launch_moddown_prep_multi_tensor_pow2)(int logn, const TensorSrc& T, const PrepMulti& M, int ndrop, int batch,
                                                 const ModDownPrep& P, const PrimeDev* primes, const TW* tw_arena,
                                                 hipStream_t st)
{
  switch (logn) {
    HX_SZ(13, launch_prep_multi_tensor<13>(T, M, ndrop, batch, P, primes, tw_arena, st))
    HX_SZ(14, launch_prep_multi_tensor<14>(T, M, ndrop, batch, P, primes, tw_arena, st))
    HX_SZ(15, launch_prep_multi_tensor<15>(T, M, ndrop, batch, P, primes, tw_arena, st))
  }
  return hipErrorInvalidValue;
}
hipError_t HX_ENTRY(launch_moddown_apply_plain_tensor_pow2)(int logn, const TensorSrc& T, const PolyBases& outs, const NttRows& keep,
                                                  int nkeep, int batch, const ModDownApply& A, const PrimeDev* primes,
                                                  const TW* tw_arena, hipStream_t st)
{
  switch (logn) {
    HX_SZ(13, launch_apply_plain_tensor<13>(T, outs, keep, nkeep, batch, A, primes, tw_arena, st))
    HX_SZ(14, launch_apply_plain_tensor<14>(T, outs, keep, nkeep, batch, A, primes, tw_arena, st))
    HX_SZ(15, launch_apply_plain_tensor<15>(T, outs, keep, nkeep, batch, A, primes, tw_arena, st))
  }
  return hipErrorInvalidValue;
}

// Several dropped primes, two launches the engine brackets around its basis-extension kernel:
//   (1) inverse transform of ONE dropped row of every listed poly (with the mod-up factor folded
//       into the last stage, as in the single-prime path) into the x block P.xs / P.poly_stride;
//   (2) forward transform of delta * P^-1 (A.delta) on every kept row with the store
//       c_r <- c_r*cf - NTT(.)  -- the apply kernel with a plain load.
template <int LOGN>
static hipError_t launch_prep_one(const PolyBases& polys, int drop_row, int drop_prime, int batch,
                                  const ModDownPrep& P, const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  hipError_t e = moddown_attrs<LOGN>();
  if (e != hipSuccess)
    return e;
  HX_LAUNCH((ntt_moddown_prep_kernel<LOGN>), dim3((unsigned)polys.n * (unsigned)batch),
                     dim3(Geo<LOGN>::T), lds_bytes, st, polys, drop_row, drop_prime, batch, P, primes,
                     tw_arena);
  return hipGetLastError();
}
template <int LOGN>
static hipError_t launch_apply_plain(const PolyBases& polys, const PolyBases& outs, const NttRows& keep,
                                     int nkeep, int batch, const ModDownApply& A, const PrimeDev* primes,
                                     const TW* tw_arena, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  hipError_t e = moddown_attrs<LOGN>();
  if (e != hipSuccess)
    return e;
  HX_LAUNCH((ntt_moddown_apply_kernel<LOGN, true>),
                     dim3(moddown_apply_grid((unsigned)polys.n, (unsigned)nkeep, (unsigned)batch)),
                     dim3(Geo<LOGN>::T), lds_bytes, st, polys, outs, keep, nkeep, batch, A, primes, tw_arena);
  return hipGetLastError();
}
template <int LOGN>
static hipError_t launch_prep_multi(const PolyBases& polys, const PrepMulti& M, int ndrop, int batch,
                                    const ModDownPrep& P, const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)Geo<LOGN>::LDS_WORDS * 4;
  hipError_t e = moddown_attrs<LOGN>();
  if (e != hipSuccess)
    return e;
  HX_LAUNCH((ntt_moddown_prep_multi_kernel<LOGN>), dim3((unsigned)ndrop * (unsigned)polys.n * (unsigned)batch),
                     dim3(Geo<LOGN>::T), lds_bytes, st, polys, M, batch, P, primes, tw_arena);
  return hipGetLastError();
}
// P.xs = x block of the first listed prime, P.poly_stride = words between the x blocks of two polys,
// x blocks of consecutive listed primes batch*N words apart
hipError_t HX_ENTRY(launch_moddown_prep_multi_pow2)(int logn, const PolyBases& polys, const PrepMulti& M, int ndrop, int batch,
                                          const ModDownPrep& P, const PrimeDev* primes, const TW* tw_arena,
                                          hipStream_t st)
{
  switch (logn) {
    HX_SZ(13, launch_prep_multi<13>(polys, M, ndrop, batch, P, primes, tw_arena, st))
    HX_SZ(14, launch_prep_multi<14>(polys, M, ndrop, batch, P, primes, tw_arena, st))
    HX_SZ(15, launch_prep_multi<15>(polys, M, ndrop, batch, P, primes, tw_arena, st))
  }
  return hipErrorInvalidValue;
}
hipError_t HX_ENTRY(launch_moddown_prep_pow2)(int logn, const PolyBases& polys, int drop_row, int drop_prime, int batch,
                                    const ModDownPrep& P, const PrimeDev* primes, const TW* tw_arena,
                                    hipStream_t st)
{
  switch (logn) {
    HX_SZ(13, launch_prep_one<13>(polys, drop_row, drop_prime, batch, P, primes, tw_arena, st))
    HX_SZ(14, launch_prep_one<14>(polys, drop_row, drop_prime, batch, P, primes, tw_arena, st))
    HX_SZ(15, launch_prep_one<15>(polys, drop_row, drop_prime, batch, P, primes, tw_arena, st))
  }
  return hipErrorInvalidValue;
}
hipError_t HX_ENTRY(launch_moddown_apply_plain_pow2)(int logn, const PolyBases& polys, const PolyBases& outs,
                                           const NttRows& keep, int nkeep, int batch, const ModDownApply& A,
                                           const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
  switch (logn) {
    HX_SZ(13, launch_apply_plain<13>(polys, outs, keep, nkeep, batch, A, primes, tw_arena, st))
    HX_SZ(14, launch_apply_plain<14>(polys, outs, keep, nkeep, batch, A, primes, tw_arena, st))
    HX_SZ(15, launch_apply_plain<15>(polys, outs, keep, nkeep, batch, A, primes, tw_arena, st))
  }
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------
// Small rings (N = 2^logn <= 4096; the reference's own unit tests use phi(m) in
// {8, 64, 256}, tests/TestHEXL.cpp:139-231): one workgroup per row, the row lives
// in LDS as 64-bit words, one __syncthreads per stage.  Not a throughput kernel.
// ---------------------------------------------------------------------
template <bool INV>
__global__ void __launch_bounds__(256)
ntt_small_kernel(const uint64_t* in, uint64_t* out, NttRows rows, int batch, int logn,
                 const PrimeDev* __restrict__ primes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint64_t sm[];
  const unsigned N = 1u << logn, tid = threadIdx.x, nth = blockDim.x;
  const unsigned ri = blockIdx.x / (unsigned)batch;
  const int b = (int)(blockIdx.x % (unsigned)batch);
  const PrimeDev* pd = primes + rows.prime[ri];
  const size_t roff = ((size_t)rows.row[ri] * batch + b) * (size_t)N;
  const TW* tw = tw_arena + (INV ? pd->tw_inv_off : pd->tw_fwd_off);
  const uint64_t q = pd->q, q2 = q + q;
  const uint64_t* src = in + roff;
  uint64_t* dst = out + roff;
  for (unsigned i = tid; i < N; i += nth)
    sm[INV ? (__brev(i) >> (32 - logn)) : i] = src[i];
  __syncthreads();
  if (!INV) {
    for (int s = 0; s < logn; s++) {
      const unsigned t = N >> (s + 1);
      for (unsigned bf = tid; bf < N / 2; bf += nth) {
        const unsigned g = bf >> (logn - 1 - s), j = bf & (t - 1), p = 2 * g * t + j;
        ct_bfly(sm[p], sm[p + t], tw[(1u << s) + g], q, q2);
      }
      __syncthreads();
    }
    for (unsigned p = tid; p < N; p += nth)
      dst[__brev(p) >> (32 - logn)] = norm4(sm[p], q, q2);
  } else {
    for (int s = logn - 1; s >= 0; s--) {
      const unsigned t = N >> (s + 1);
      for (unsigned bf = tid; bf < N / 2; bf += nth) {
        const unsigned g = bf >> (logn - 1 - s), j = bf & (t - 1), p = 2 * g * t + j;
        gs_bfly(sm[p], sm[p + t], tw[(1u << s) + g], q, q2);
      }
      __syncthreads();
    }
    const TW ninv = tw[0];
    for (unsigned i = tid; i < N; i += nth)
      dst[i] = norm2(shoup_lazy(sm[i], ninv, q), q);
  }
}

static hipError_t launch_small(bool inverse, const uint64_t* in, uint64_t* out, const NttRows& rows,
                               int nrows, int batch, int logn, const PrimeDev* primes,
                               const TW* tw_arena, hipStream_t st)
{
  unsigned N = 1u << logn;
  unsigned threads = N / 2 < 64 ? 64 : (N / 2 > 256 ? 256 : N / 2);
  dim3 grid((unsigned)nrows * (unsigned)batch), block(threads);
  size_t lds = (size_t)N * 8;
  if (inverse)
    HX_LAUNCH(ntt_small_kernel<true>, grid, block, lds, st, in, out, rows, batch, logn,
                       primes, tw_arena);
  else
    HX_LAUNCH(ntt_small_kernel<false>, grid, block, lds, st, in, out, rows, batch, logn,
                       primes, tw_arena);
  return hipGetLastError();
}

#if !defined(HX_NTT_ONLY) || HX_NTT_ONLY == 14
// =====================================================================
// The FORWARD transform of a 2^15-point row as two independent 2^14-point workgroups (Cmodulus::FFT,
// src/CModulus.cpp:389-426, at the CKKS chain's ring m = 65536).  ntt_row_kernel<15> is one 1024-thread workgroup per
// row and CU -- nothing covers its load and barrier phases: 5.3 ps per coefficient where the 2^14-point kernel, two
// 512-thread workgroups per CU, takes 4.2.  The first Cooley-Tukey stage pairs x[p] with x[p + N/2] under ONE twiddle
// T = psi^(N/2): workgroup g = 0 / 1 forms x[p] +/- T x[p + N/2] in its load and continues with the sub-transform's own
// tables (build_tw_tables_sub, OUT = 1: what the split power-of-two rings use), and its outputs are the row's
// evaluations 2 j + g (big_post's interleave for S = 2).  The inverse needs the cross-half stage LAST, behind both
// sub-transforms: it stays on the one-workgroup kernel.
// MEASURED SLOWER (round 6, config 4, same box): 818-822 us against 783-798 for the 4608 rows of a multiply of the
// batch (1032 with non-temporal row IO: the halves fill each other's cache lines) -- each half reads the whole row and
// stores every other word.  Off by default (HX_HALF15=1); profiles/r06_ab_half_row_forward_2p15.json.
// =====================================================================
template <int LB, bool PROTH>
#ifndef HX_HALF_AUX
#define HX_HALF_AUX 0   // (both workgroups of a row read all of it and fill each other's cache lines: not non-temporal)
#endif
struct HalfIO15 {
  static constexpr unsigned Q = 1u << 14;
  // bound of what load() hands over, in q: Proth-form rows start below 2q (canonical, or the digit kernel's Montgomery
  // sums: BufIOT) -> x + R < 2q + q (1 + 2/16 + 2^-32), 2x + 2q - (x + R) < 4q; other rows (shoup4: R < 4q) x + R and
  // x + 4q - R below 8q + 4q resp. q + 4q
  static constexpr int LOAD_BOUND = PROTH ? 4 : (LB == 8 ? 12 : 5);
  static constexpr bool LAZY_STORE = false;
  static constexpr bool PIPELINED = false;
  struct StorePrefetch {};
  v4i32 rin, rout;
  QC qc;
  TW t1;
  TWM t1m;
  unsigned g;
  __device__ HalfIO15(const uint64_t* in_row, uint64_t* out_row, const PrimeDev* pd, unsigned g_)
      : rin(make_rsrc(in_row, 2u * Q * 8u)), rout(make_rsrc(out_row, 2u * Q * 8u)), qc(make_qc(pd->q, pd->mu64)), t1m(pd->half_t1m), g(g_)
  {
    t1.w = pd->half_t1;
    t1.wp = pd->half_t1p;
  }
  __device__ __forceinline__ uint64_t load(unsigned tid, unsigned c) const
  {
    const v2i32 a = hx_buffer_load_v2(rin, (int)(tid * 8u), (int)(c * 8u), HX_HALF_AUX);
    const v2i32 b = hx_buffer_load_v2(rin, (int)(tid * 8u), (int)((c + Q) * 8u), HX_HALF_AUX);
    const uint64_t x = ((uint64_t)(uint32_t)a.y << 32) | (uint32_t)a.x, y = ((uint64_t)(uint32_t)b.y << 32) | (uint32_t)b.x;
    if constexpr (PROTH) {
      const uint64_t xn = mont_acc(y, t1m, qc, x);           // x + T y
      return g ? (x << 1) + qc.q2 - xn : xn;                  // x - T y + 2q
    } else {
      const uint64_t r = shoup4(y, t1, 0 - qc.q);            // [0, 4q)
      return g ? x + (qc.q2 << 1) - r : x + r;
    }
  }
  __device__ __forceinline__ void store(unsigned tid, unsigned c, uint64_t v) const
  {
    v2i32 d;
    d.x = (int)(uint32_t)v;
    d.y = (int)(uint32_t)(v >> 32);
    hx_buffer_store_v2(d, rout, (int)(tid * 16u), (int)(c * 16u + g * 8u), HX_HALF_AUX);
  }
  __device__ __forceinline__ TW last_tw(TW def, int) const { return def; }
  __device__ __forceinline__ TWM last_tw(TWM def, int) const { return def; }
};

// rows.prime[i] = index (into `primes`: the context's sub-transform table) of row i's entry for g = 0; g = 1 follows it
template <int LB>
__global__ void __launch_bounds__(Geo<14>::T, HX_NTT_MINWAVES(14))
ntt_row_half15_kernel(const uint64_t* in, uint64_t* out, NttRows rows, int batch,
                      const PrimeDev* __restrict__ primes, const TW* __restrict__ tw_arena)
{
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned g = wid & 1u, rb = wid >> 1;            // the two halves of a row side by side (same XCD: one L2 sees both)
  const unsigned ri = rb / (unsigned)batch;
  const int b = (int)(rb % (unsigned)batch);
  const int row = (int)uniform_u16(rows.row, ri);
  const PrimeDev* pd = primes + uniform_u16(rows.prime, ri) + g;
  const size_t roff = ((size_t)row * batch + b) * (size_t)(2u << 14);
  const TW* tw = tw_arena + pd->tw_fwd_off;
  const QC q = make_qc(pd->q, pd->mu64);
#ifndef HX_NO_PROTH
  if (pd->proth) {
    const HalfIO15<LB, true> io(in + roff, out + roff, pd, g);
    ntt_body_ar<14, false, ArProth>(lds, io, reinterpret_cast<const TWM*>(tw), q);
    return;
  }
#endif
  const HalfIO15<LB, false> io(in + roff, out + roff, pd, g);
  ntt_body_ar<14, false, ArShoup>(lds, io, tw, q);
}
template <int LB>
static hipError_t launch_half15(const uint64_t* in, uint64_t* out, const NttRows& rows, int nrows, int batch,
                                const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)Geo<14>::LDS_WORDS * 4;
  hipError_t e = hxp::dyn_lds((const void*)ntt_row_half15_kernel<LB>, (int)lds_bytes);
  if (e != hipSuccess)
    return e;
  HX_LAUNCH((ntt_row_half15_kernel<LB>), dim3(2u * (unsigned)nrows * (unsigned)batch), dim3(Geo<14>::T), lds_bytes, st, in, out,
            rows, batch, primes, tw_arena);
  return hipGetLastError();
}
// in != out (the two workgroups of a row read all of it); lazy_in: the words are in [0,8q) (Proth-form rows: below 2q)
hipError_t launch_ntt_half15_fwd(bool lazy_in, const uint64_t* in, uint64_t* out, const NttRows& rows, int nrows, int batch,
                                 const PrimeDev* sub_primes, const TW* tw_arena, hipStream_t st)
{
  return lazy_in ? launch_half15<8>(in, out, rows, nrows, batch, sub_primes, tw_arena, st)
                 : launch_half15<1>(in, out, rows, nrows, batch, sub_primes, tw_arena, st);
}
#endif

// forward transform of rows whose words are lazy, in [0,8q) (N = 2^13..2^15 only: ntt_lazy_input_ok)
hipError_t HX_ENTRY(launch_ntt_pow2_lazy_in)(int logn, const uint64_t* in, uint64_t* out, const NttRows& rows, int nrows, int batch,
                                   const PrimeDev* primes, const TW* tw_arena, hipStream_t st)
{
  switch (logn) {
    HX_SZ(13, launch_one<13, false, 8>(in, out, rows, nrows, batch, primes, tw_arena, st))
    HX_SZ(14, launch_one<14, false, 8>(in, out, rows, nrows, batch, primes, tw_arena, st))
    HX_SZ(15, launch_one<15, false, 8>(in, out, rows, nrows, batch, primes, tw_arena, st))
  }
  return hipErrorInvalidValue;
}

// entry point used by engine.hip: transform `nrows` (<= MAX_ROWS) listed rows, in -> out
// (in == out allowed: a workgroup reads its whole row before it writes it).
hipError_t HX_ENTRY(launch_ntt_pow2)(int logn, bool inverse, const uint64_t* in, uint64_t* out,
                           const NttRows& rows, int nrows, int batch, const PrimeDev* primes,
                           const TW* tw_arena, hipStream_t st)
{
  if (logn >= 1 && logn <= 12)
    return launch_small(inverse, in, out, rows, nrows, batch, logn, primes, tw_arena, st);
  switch (logn) {
    HX_SZ(13, inverse ? launch_one<13, true>(in, out, rows, nrows, batch, primes, tw_arena, st)
                      : launch_one<13, false>(in, out, rows, nrows, batch, primes, tw_arena, st))
    HX_SZ(14, inverse ? launch_one<14, true>(in, out, rows, nrows, batch, primes, tw_arena, st)
                      : launch_one<14, false>(in, out, rows, nrows, batch, primes, tw_arena, st))
    HX_SZ(15, inverse ? launch_one<15, true>(in, out, rows, nrows, batch, primes, tw_arena, st)
                      : launch_one<15, false>(in, out, rows, nrows, batch, primes, tw_arena, st))
  }
  return hipErrorInvalidValue;
}

}  // namespace hx

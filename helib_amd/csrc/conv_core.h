// conv_core.h -- host/device pieces of the long negacyclic convolutions used by the
// Bluestein transform (general m; HElib src/bluestein.cpp:134-201 calls NTL's fftRep
// TofftRep/mul/FromfftRep for these).  A 2^k-point negacyclic product is an exact LINEAR
// convolution whenever deg(a)+deg(b) < 2^k, which is how it is used here.
//
// Sizes up to 2^15 run directly on the row kernels of ntt_core.h.  Sizes 2^16 / 2^17 are
// split radix-4 (2^18 radix-8, 2^19 radix-16: below): the first two Cooley-Tukey stages are done by split_fwd4 on four elements
// Q = 2^(k-2) apart, then four independent Q-point sub-transforms follow, each with its own
// twiddle table (build_tw_tables_sub, OUT = 2).  The output order of the split transform is
// not the natural one -- irrelevant for convolutions (forward, pointwise, inverse).
#pragma once
#include "ntt_core.h"

namespace hx {

HXD uint64_t shoup_full(uint64_t x, TW t, uint64_t q) { return norm2(shoup_lazy(x, t, q), q); }
HXD uint64_t addm(uint64_t a, uint64_t b, uint64_t q)
{
  uint64_t s = a + b;
  return s >= q ? s - q : s;
}
HXD uint64_t subm(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }

// constants of one split (per prime): T[1..3] = psi_rev_full[1..3]; inverse side:
// iT2, iT3 = their inverses, iT1q = T1^-1 / 4, quarter = 1/4.
struct SplitTW {
  TW T1, T2, T3;
  TW iT1q, iT2, iT3, quarter;
};

// inputs canonical (or 0), outputs canonical; b[g] goes to sub-transform g
HXD void split_fwd4(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t a3, const SplitTW& S, uint64_t q,
                    uint64_t (&b)[4])
{
  uint64_t t2 = shoup_full(a2, S.T1, q), t3 = shoup_full(a3, S.T1, q);
  uint64_t e0 = addm(a0, t2, q), e2 = subm(a0, t2, q);
  uint64_t e1 = addm(a1, t3, q), e3 = subm(a1, t3, q);
  uint64_t u1 = shoup_full(e1, S.T2, q), u3 = shoup_full(e3, S.T3, q);
  b[0] = addm(e0, u1, q);
  b[1] = subm(e0, u1, q);
  b[2] = addm(e2, u3, q);
  b[3] = subm(e2, u3, q);
}
// c[g] = output of sub-inverse g (already scaled by 1/Q); a[] = the four time-domain values
HXD void split_inv4(const uint64_t (&c)[4], const SplitTW& S, uint64_t q, uint64_t (&a)[4])
{
  uint64_t e0 = addm(c[0], c[1], q), e1 = shoup_full(subm(c[0], c[1], q), S.iT2, q);
  uint64_t e2 = addm(c[2], c[3], q), e3 = shoup_full(subm(c[2], c[3], q), S.iT3, q);
  a[0] = shoup_full(addm(e0, e2, q), S.quarter, q);
  a[2] = shoup_full(subm(e0, e2, q), S.iT1q, q);
  a[1] = shoup_full(addm(e1, e3, q), S.quarter, q);
  a[3] = shoup_full(subm(e1, e3, q), S.iT1q, q);
}

// ---- radix-2^LS split, LS = 3 (convolution size 2^18 = 8 x 2^15) or 4 (2^19 = 16 x 2^15): the first
// LS Cooley-Tukey stages on 2^LS elements Q = 2^(k-LS) apart, generic form of the above.
// T[idx] = psi_rev_full[idx] for idx = 1 .. 2^LS - 1 (stage s uses idx = 2^s + group), iT[idx] their
// inverses; the inverse folds 1/2^LS into its last stage (iT1e = T[1]^-1 / 2^LS, inv = 1/2^LS).
template <int LS>
struct SplitTWN {
  TW T[1 << LS], iT[1 << LS];
  TW iT1e, inv;
};
typedef SplitTWN<3> SplitTW8;
typedef SplitTWN<4> SplitTW16;
template <int LS>
HXD void split_fwdN(uint64_t (&e)[1 << LS], const SplitTWN<LS>& S, uint64_t q)
{
  constexpr int R = 1 << LS;
  for (int s = 0; s < LS; s++) {
    const int m = 1 << s, half = (R / 2) >> s;
    for (int g = 0; g < m; g++)
      for (int j = 0; j < half; j++) {
        const int a = g * 2 * half + j, b = a + half;
        const uint64_t t = shoup_full(e[b], S.T[m + g], q);
        const uint64_t x = e[a];
        e[a] = addm(x, t, q);
        e[b] = subm(x, t, q);
      }
  }
}
template <int LS>
HXD void split_invN(uint64_t (&e)[1 << LS], const SplitTWN<LS>& S, uint64_t q)
{
  constexpr int R = 1 << LS;
  for (int s = LS - 1; s >= 1; s--) {
    const int m = 1 << s, half = (R / 2) >> s;
    for (int g = 0; g < m; g++)
      for (int j = 0; j < half; j++) {
        const int a = g * 2 * half + j, b = a + half;
        const uint64_t x = e[a], y = e[b];
        e[a] = addm(x, y, q);
        e[b] = shoup_full(subm(x, y, q), S.iT[m + g], q);
      }
  }
  for (int j = 0; j < R / 2; j++) {
    const uint64_t x = e[j], y = e[j + R / 2];
    e[j] = shoup_full(addm(x, y, q), S.inv, q);
    e[j + R / 2] = shoup_full(subm(x, y, q), S.iT1e, q);
  }
}
HXD void split_fwd8(uint64_t (&e)[8], const SplitTW8& S, uint64_t q) { split_fwdN<3>(e, S, q); }
HXD void split_inv8(uint64_t (&e)[8], const SplitTW8& S, uint64_t q) { split_invN<3>(e, S, q); }


// value i of the reconstructed 4Q-point result from the four sub-block outputs c[g] at position i mod Q
// (one output of split_inv4)
HXD uint64_t split_inv4_one(const uint64_t (&c)[4], const SplitTW& S, uint64_t q, unsigned quarter)
{
  if ((quarter & 1u) == 0) {
    const uint64_t e0 = addm(c[0], c[1], q), e2 = addm(c[2], c[3], q);
    return quarter == 0 ? shoup_full(addm(e0, e2, q), S.quarter, q) : shoup_full(subm(e0, e2, q), S.iT1q, q);
  }
  const uint64_t e1 = shoup_full(subm(c[0], c[1], q), S.iT2, q), e3 = shoup_full(subm(c[2], c[3], q), S.iT3, q);
  return quarter == 1 ? shoup_full(addm(e1, e3, q), S.quarter, q) : shoup_full(subm(e1, e3, q), S.iT1q, q);
}

}  // namespace hx

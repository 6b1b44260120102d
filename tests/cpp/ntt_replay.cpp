// CPU replay of the HIP NTT kernel's phase functions (helib_amd/csrc/ntt_core.h)
// thread-by-thread, so that the index arithmetic, twiddle-table layout and lazy
// reduction bounds of the kernel can be checked against the oracle without a GPU.
// TEST INFRASTRUCTURE: built by tests/, never linked into the product library.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../helib_amd/csrc/ntt_core.h"

typedef unsigned __int128 u128;
static uint64_t mm(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
static uint64_t pw(uint64_t a, uint64_t e, uint64_t q)
{
  uint64_t r = 1;
  while (e) {
    if (e & 1) r = mm(r, a, q);
    a = mm(a, a, q);
    e >>= 1;
  }
  return r;
}

// lazy-input forward rows (BufIOT<8> on the device): the bound the exact-RNS kernels hand over
template <int LB>
struct PtrIOLB : hx::PtrIO {
  static constexpr int LOAD_BOUND = LB;
};
// the digit rows as the device reads them (BufIOT<8>): declared bound 8, rows of Proth-form primes at bound 2
struct PtrIOLB8P2 : hx::PtrIO {
  static constexpr int LOAD_BOUND = 8;
  template <class AR>
  static constexpr int load_bound() { return AR::PROTH ? 2 : 8; }
};
template <int LOGN, int PH, class AR, class IO>
static void run_phase(bool inverse, std::vector<uint64_t>& V, std::vector<uint32_t>& NL,
                      std::vector<uint32_t>& lds, const uint64_t* in, uint64_t* out,
                      const typename AR::Tw* tw, uint64_t q)
{
  using R = hx::RowNTT<LOGN, AR>;
  constexpr int T = hx::Geo<LOGN>::T;
  for (unsigned tid = 0; tid < (unsigned)T; tid++) {
    uint64_t(&v)[32] = *reinterpret_cast<uint64_t(*)[32]>(&V[tid * 32]);
    uint32_t(&nl)[32] = *reinterpret_cast<uint32_t(*)[32]>(&NL[tid * 32]);
    IO io{{in, out}};
    if (inverse)
      R::template inv<PH>(tid, v, nl, lds.data(), io, tw, hx::make_qc(q));
    else
      R::template fwd<PH>(tid, v, nl, lds.data(), io, tw, hx::make_qc(q));
  }
}

template <class AR>
static std::vector<typename AR::Tw> table_of(const std::vector<hx::TW>& t, uint64_t q);
template <>
std::vector<hx::TW> table_of<hx::ArShoup>(const std::vector<hx::TW>& t, uint64_t) { return t; }
template <>
std::vector<hx::TWM> table_of<hx::ArProth>(const std::vector<hx::TW>& t, uint64_t q)
{
  std::vector<hx::TWM> o(t.size());
  hx::tw_tables_to_mont(t.data(), (int)t.size(), q, o.data());
  return o;
}

template <int LOGN, class AR, class IO = PtrIOLB<1>>
static int replay(int inverse, uint64_t q, uint64_t psi, const uint64_t* in, uint64_t* out)
{
  using G = hx::Geo<LOGN>;
  std::vector<hx::TW> f(G::TW_TOTAL), i(G::TW_TOTAL);
  uint64_t psi_inv = pw(psi, q - 2, q);
  uint64_t n_inv = pw((uint64_t)G::N % q, q - 2, q);
  hx::build_tw_tables<LOGN>(q, psi, psi_inv, n_inv, mm, f.data(), i.data());
  const std::vector<typename AR::Tw> tab = table_of<AR>(inverse ? i : f, q);
  std::vector<uint64_t> V((size_t)G::T * 32);
  std::vector<uint32_t> NL((size_t)G::T * 32);
  std::vector<uint32_t> lds(G::LDS_WORDS, 0xdeadbeef);
  std::vector<uint64_t> inc(in, in + G::N);  // allow in == out
  const typename AR::Tw* tw = tab.data();
  run_phase<LOGN, 0, AR, IO>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 1, AR, IO>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 2, AR, IO>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 3, AR, IO>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 4, AR, IO>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 5, AR, IO>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 6, AR, IO>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 7, AR, IO>(inverse, V, NL, lds, inc.data(), out, tw, q);
  return 0;
}

extern "C" int ntt_replay(int logn, int inverse, uint64_t q, uint64_t psi, const uint64_t* in,
                          uint64_t* out)
{
  switch (logn) {
    case 13: return replay<13, hx::ArShoup>(inverse, q, psi, in, out);
    case 14: return replay<14, hx::ArShoup>(inverse, q, psi, in, out);
    case 15: return replay<15, hx::ArShoup>(inverse, q, psi, in, out);
  }
  return -1;
}
// the Proth-form arithmetic (q = 1 mod 2^32): -2 when q is not of that form.  lazy8 != 0: forward transform of
// words in [0,8q) (the load bound of the rows the exact-RNS kernels leave unreduced)
extern "C" int ntt_replay_proth(int logn, int inverse, int lazy8, uint64_t q, uint64_t psi, const uint64_t* in,
                                uint64_t* out)
{
  if (!hx::is_proth32(q))
    return -2;
  if (lazy8 == 2 && !inverse)   // words in [0,2q): the digit kernel's Proth-form target sums
    switch (logn) {
      case 13: return replay<13, hx::ArProth, PtrIOLB8P2>(0, q, psi, in, out);
      case 14: return replay<14, hx::ArProth, PtrIOLB8P2>(0, q, psi, in, out);
      case 15: return replay<15, hx::ArProth, PtrIOLB8P2>(0, q, psi, in, out);
    }
  if (lazy8 && !inverse)
    switch (logn) {
      case 13: return replay<13, hx::ArProth, PtrIOLB<8>>(0, q, psi, in, out);
      case 14: return replay<14, hx::ArProth, PtrIOLB<8>>(0, q, psi, in, out);
      case 15: return replay<15, hx::ArProth, PtrIOLB<8>>(0, q, psi, in, out);
    }
  switch (logn) {
    case 13: return replay<13, hx::ArProth>(inverse, q, psi, in, out);
    case 14: return replay<14, hx::ArProth>(inverse, q, psi, in, out);
    case 15: return replay<15, hx::ArProth>(inverse, q, psi, in, out);
  }
  return -1;
}
// The inverse transform of a pointwise product formed on load (MulLoadIO<true>, ntt_kernels.hip: the s^2 part of a tensor
// product in hx_mul_relin): the 128-bit product reduced by mont_redc128 alone, handed to the Proth-form inverse passes at
// bound 2, its 2^-64 given back by the last stage's constants times 2^128.  Same arithmetic as the device functor, with
// plain pointers for its buffer loads; every lazy bound asserted (HX_CHECK_BOUNDS).
struct MulPtrIO {
  static constexpr int LOAD_BOUND = 1;
  static constexpr bool LAZY_STORE = false;
  static constexpr bool PIPELINED = false;
  template <class AR>
  static constexpr int inv_load_bound() { return 2; }
  struct StorePrefetch {};
  const uint64_t* a;
  const uint64_t* b;
  uint64_t* out;
  uint64_t q, r2;
  uint64_t load(unsigned tid, unsigned c) const
  {
    const u128 p = (u128)a[tid + c] * b[tid + c];
    return hx::mont_redc128((uint64_t)p, (uint64_t)(p >> 64), hx::make_qc(q, 0));
  }
  void store(unsigned tid, unsigned c, uint64_t v) const { out[tid + c] = v; }
  hx::TW last_tw(hx::TW def, int) const { return def; }
  hx::TWM last_tw(hx::TWM def, int) const
  {
    const hx::QC qc = hx::make_qc(q, 0);
    return hx::csub(hx::mont_mul(def, r2, qc), q);
  }
};
template <int LOGN, int PH>
static void run_mul_phase(std::vector<uint64_t>& V, std::vector<uint32_t>& NL, std::vector<uint32_t>& lds, const MulPtrIO& io,
                          const hx::TWM* tw, uint64_t q)
{
  using R = hx::RowNTT<LOGN, hx::ArProth>;
  constexpr int T = hx::Geo<LOGN>::T;
  for (unsigned tid = 0; tid < (unsigned)T; tid++) {
    uint64_t(&v)[32] = *reinterpret_cast<uint64_t(*)[32]>(&V[tid * 32]);
    uint32_t(&nl)[32] = *reinterpret_cast<uint32_t(*)[32]>(&NL[tid * 32]);
    R::template inv<PH>(tid, v, nl, lds.data(), io, tw, hx::make_qc(q));
  }
}
template <int LOGN>
static int replay_mul(uint64_t q, uint64_t psi, const uint64_t* a, const uint64_t* b, uint64_t* out)
{
  using G = hx::Geo<LOGN>;
  std::vector<hx::TW> f(G::TW_TOTAL), i(G::TW_TOTAL);
  uint64_t psi_inv = pw(psi, q - 2, q);
  uint64_t n_inv = pw((uint64_t)G::N % q, q - 2, q);
  hx::build_tw_tables<LOGN>(q, psi, psi_inv, n_inv, mm, f.data(), i.data());
  const std::vector<hx::TWM> tab = table_of<hx::ArProth>(i, q);
  std::vector<uint64_t> V((size_t)G::T * 32);
  std::vector<uint32_t> NL((size_t)G::T * 32);
  std::vector<uint32_t> lds(G::LDS_WORDS, 0xdeadbeef);
  std::vector<uint64_t> ac(a, a + G::N), bc(b, b + G::N);
  const uint64_t r1 = (uint64_t)((((u128)1) << 64) % q);
  const MulPtrIO io{ac.data(), bc.data(), out, q, mm(r1, r1, q)};
  run_mul_phase<LOGN, 0>(V, NL, lds, io, tab.data(), q);
  run_mul_phase<LOGN, 1>(V, NL, lds, io, tab.data(), q);
  run_mul_phase<LOGN, 2>(V, NL, lds, io, tab.data(), q);
  run_mul_phase<LOGN, 3>(V, NL, lds, io, tab.data(), q);
  run_mul_phase<LOGN, 4>(V, NL, lds, io, tab.data(), q);
  run_mul_phase<LOGN, 5>(V, NL, lds, io, tab.data(), q);
  run_mul_phase<LOGN, 6>(V, NL, lds, io, tab.data(), q);
  run_mul_phase<LOGN, 7>(V, NL, lds, io, tab.data(), q);
  return 0;
}
extern "C" int ntt_replay_proth_mul(int logn, uint64_t q, uint64_t psi, const uint64_t* a, const uint64_t* b, uint64_t* out)
{
  if (!hx::is_proth32(q))
    return -2;
  switch (logn) {
    case 13: return replay_mul<13>(q, psi, a, b, out);
    case 14: return replay_mul<14>(q, psi, a, b, out);
    case 15: return replay_mul<15>(q, psi, a, b, out);
  }
  return -1;
}
// x + R of mont_acc for n (y, W, x) triples: the word-wise arithmetic the device runs, for the python-integer
// restatement in tests/test_host_logic.py
extern "C" void mont_acc_replay(uint64_t q, long n, const uint64_t* y, const uint64_t* W, const uint64_t* x, uint64_t* out)
{
  const hx::QC c = hx::make_qc(q);
  for (long i = 0; i < n; i++)
    out[i] = hx::mont_acc(y[i], W[i], c, x[i]);
}

// ---------------------------------------------------------------------------------------
// CPU replay of the radix-4 split convolution (conv_core.h): c = a * b mod (X^(4Q) + 1) through
// split_fwd4 + four Q-point sub-transforms (replayed kernel phases with build_tw_tables_sub
// tables) + pointwise + inverse.  Checks the split math and the sub-transform twiddle tables.
// ---------------------------------------------------------------------------------------
#include "../../helib_amd/csrc/conv_core.h"

template <int LOGQ>
static void sub_transform(bool inverse, const hx::TW* tw, uint64_t q, const uint64_t* in, uint64_t* out)
{
  using G = hx::Geo<LOGQ>;
  std::vector<uint64_t> V((size_t)G::T * 32);
  std::vector<uint32_t> NL((size_t)G::T * 32);
  std::vector<uint32_t> lds(G::LDS_WORDS, 0);
  std::vector<uint64_t> inc(in, in + G::N);
  run_phase<LOGQ, 0, hx::ArShoup, PtrIOLB<1>>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGQ, 1, hx::ArShoup, PtrIOLB<1>>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGQ, 2, hx::ArShoup, PtrIOLB<1>>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGQ, 3, hx::ArShoup, PtrIOLB<1>>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGQ, 4, hx::ArShoup, PtrIOLB<1>>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGQ, 5, hx::ArShoup, PtrIOLB<1>>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGQ, 6, hx::ArShoup, PtrIOLB<1>>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGQ, 7, hx::ArShoup, PtrIOLB<1>>(inverse, V, NL, lds, inc.data(), out, tw, q);
}

template <int LOGQ>
static int split_conv(uint64_t q, uint64_t psi /* primitive 2^(LOGQ+3)-th root */, const uint64_t* a,
                      const uint64_t* b, uint64_t* c)
{
  using G = hx::Geo<LOGQ>;
  const int Q = G::N, FULL = LOGQ + 2;
  uint64_t psi_inv = pw(psi, q - 2, q);
  uint64_t qinv = pw((uint64_t)Q % q, q - 2, q);
  std::vector<std::vector<hx::TW>> F(4, std::vector<hx::TW>(G::TW_TOTAL)), I(4, std::vector<hx::TW>(G::TW_TOTAL));
  for (unsigned g = 0; g < 4; g++)
    hx::build_tw_tables_sub<LOGQ>(q, psi, psi_inv, qinv, mm, 2, g, F[g].data(), I[g].data());
  auto mk = [&](uint64_t w) { hx::TW t; t.w = w; t.wp = (uint64_t)((((u128)w) << 64) / q); return t; };
  auto prev = [&](unsigned idx) { return pw(psi, hx::brev_bits(idx, FULL), q); };
  hx::SplitTW S;
  uint64_t T1 = prev(1), T2 = prev(2), T3 = prev(3), quarter = pw(4, q - 2, q);
  S.T1 = mk(T1); S.T2 = mk(T2); S.T3 = mk(T3);
  S.iT2 = mk(pw(T2, q - 2, q)); S.iT3 = mk(pw(T3, q - 2, q));
  S.iT1q = mk(mm(pw(T1, q - 2, q), quarter, q)); S.quarter = mk(quarter);
  std::vector<uint64_t> qa(4 * Q), qb(4 * Q), fa(4 * Q), fb(4 * Q);
  for (int p = 0; p < Q; p++) {
    uint64_t o[4];
    hx::split_fwd4(a[p], a[p + Q], a[p + 2 * Q], a[p + 3 * Q], S, q, o);
    for (int g = 0; g < 4; g++) qa[g * Q + p] = o[g];
    hx::split_fwd4(b[p], b[p + Q], b[p + 2 * Q], b[p + 3 * Q], S, q, o);
    for (int g = 0; g < 4; g++) qb[g * Q + p] = o[g];
  }
  for (int g = 0; g < 4; g++) {
    sub_transform<LOGQ>(false, F[g].data(), q, qa.data() + g * Q, fa.data() + g * Q);
    sub_transform<LOGQ>(false, F[g].data(), q, qb.data() + g * Q, fb.data() + g * Q);
    for (int j = 0; j < Q; j++) fa[g * Q + j] = mm(fa[g * Q + j], fb[g * Q + j], q);
    sub_transform<LOGQ>(true, I[g].data(), q, fa.data() + g * Q, qa.data() + g * Q);
  }
  for (int p = 0; p < Q; p++) {
    uint64_t in4[4] = {qa[p], qa[Q + p], qa[2 * Q + p], qa[3 * Q + p]}, o[4];
    hx::split_inv4(in4, S, q, o);
    for (int g = 0; g < 4; g++) c[g * Q + p] = o[g];
  }
  return 0;
}

extern "C" int split_conv_replay(int logq, uint64_t q, uint64_t psi, const uint64_t* a, const uint64_t* b,
                                 uint64_t* c)
{
  switch (logq) {
    case 13: return split_conv<13>(q, psi, a, b, c);
    case 14: return split_conv<14>(q, psi, a, b, c);
    case 15: return split_conv<15>(q, psi, a, b, c);
  }
  return -1;
}


// The same for the radix-8 split (conv size 8 * 2^LOGQ): split_fwd8 + eight sub-transforms with
// build_tw_tables_sub(OUT = 3) tables + pointwise + inverse.
template <int LOGQ>
static int split_conv8(uint64_t q, uint64_t psi /* primitive 2^(LOGQ+4)-th root */, const uint64_t* a,
                       const uint64_t* b, uint64_t* c)
{
  using G = hx::Geo<LOGQ>;
  const int Q = G::N, FULL = LOGQ + 3;
  uint64_t psi_inv = pw(psi, q - 2, q);
  uint64_t qinv = pw((uint64_t)Q % q, q - 2, q);
  std::vector<std::vector<hx::TW>> F(8, std::vector<hx::TW>(G::TW_TOTAL)), I(8, std::vector<hx::TW>(G::TW_TOTAL));
  for (unsigned g = 0; g < 8; g++)
    hx::build_tw_tables_sub<LOGQ>(q, psi, psi_inv, qinv, mm, 3, g, F[g].data(), I[g].data());
  auto mk = [&](uint64_t w) { hx::TW t; t.w = w; t.wp = (uint64_t)((((u128)w) << 64) / q); return t; };
  hx::SplitTW8 S;
  uint64_t eighth = pw(8, q - 2, q);
  S.T[0] = S.iT[0] = mk(0);
  for (unsigned idx = 1; idx < 8; idx++) {
    uint64_t T = pw(psi, hx::brev_bits(idx, FULL), q);
    S.T[idx] = mk(T);
    S.iT[idx] = mk(pw(T, q - 2, q));
  }
  S.inv = mk(eighth);
  S.iT1e = mk(mm(S.iT[1].w, eighth, q));
  std::vector<uint64_t> qa(8 * Q), qb(8 * Q), fa(8 * Q), fb(8 * Q);
  for (int p = 0; p < Q; p++) {
    uint64_t e[8];
    for (int g = 0; g < 8; g++) e[g] = a[p + g * Q];
    hx::split_fwd8(e, S, q);
    for (int g = 0; g < 8; g++) qa[g * Q + p] = e[g];
    for (int g = 0; g < 8; g++) e[g] = b[p + g * Q];
    hx::split_fwd8(e, S, q);
    for (int g = 0; g < 8; g++) qb[g * Q + p] = e[g];
  }
  for (int g = 0; g < 8; g++) {
    sub_transform<LOGQ>(false, F[g].data(), q, qa.data() + g * Q, fa.data() + g * Q);
    sub_transform<LOGQ>(false, F[g].data(), q, qb.data() + g * Q, fb.data() + g * Q);
    for (int j = 0; j < Q; j++) fa[g * Q + j] = mm(fa[g * Q + j], fb[g * Q + j], q);
    sub_transform<LOGQ>(true, I[g].data(), q, fa.data() + g * Q, qa.data() + g * Q);
  }
  for (int p = 0; p < Q; p++) {
    uint64_t e[8];
    for (int g = 0; g < 8; g++) e[g] = qa[g * Q + p];
    hx::split_inv8(e, S, q);
    for (int g = 0; g < 8; g++) c[g * Q + p] = e[g];
  }
  return 0;
}

extern "C" int split_conv8_replay(int logq, uint64_t q, uint64_t psi, const uint64_t* a, const uint64_t* b,
                                  uint64_t* c)
{
  switch (logq) {
    case 13: return split_conv8<13>(q, psi, a, b, c);
    case 15: return split_conv8<15>(q, psi, a, b, c);
  }
  return -1;
}


// ---------------------------------------------------------------------------------------
// CPU replay of the power-of-two transform beyond one row kernel (N = S * 2^LOGQ, S = 4 / 8;
// engine.hip pow2_big_rows, bluestein.h big_pre / big_post): first log2(S) Cooley-Tukey stages on
// elements Q apart, S sub-transforms with their own tables, and the interleave that restores the
// natural order -- out[j] = in(psi^(2j+1)) as Cmodulus::FFT defines it -- and the way back.
// ---------------------------------------------------------------------------------------
template <int LOGQ, int S>
static int big_ntt(int inverse, uint64_t q, uint64_t psi, const uint64_t* in, uint64_t* out)
{
  using G = hx::Geo<LOGQ>;
  constexpr int LS = S == 16 ? 4 : (S == 8 ? 3 : 2);
  const int Q = G::N, FULL = LOGQ + LS;
  uint64_t psi_inv = pw(psi, q - 2, q);
  uint64_t qinv = pw((uint64_t)Q % q, q - 2, q);
  std::vector<std::vector<hx::TW>> F(S, std::vector<hx::TW>(G::TW_TOTAL)), I(S, std::vector<hx::TW>(G::TW_TOTAL));
  for (unsigned g = 0; g < (unsigned)S; g++)
    hx::build_tw_tables_sub<LOGQ>(q, psi, psi_inv, qinv, mm, LS, g, F[g].data(), I[g].data());
  auto mk = [&](uint64_t w) { hx::TW t; t.w = w; t.wp = (uint64_t)((((u128)w) << 64) / q); return t; };
  auto prev = [&](unsigned idx) { return pw(psi, hx::brev_bits(idx, FULL), q); };
  hx::SplitTW S4;
  hx::SplitTWN<(S >= 8 ? LS : 3)> S8;
  if (S == 4) {
    uint64_t T1 = prev(1), T2 = prev(2), T3 = prev(3), quarter = pw(4, q - 2, q);
    S4.T1 = mk(T1); S4.T2 = mk(T2); S4.T3 = mk(T3);
    S4.iT2 = mk(pw(T2, q - 2, q)); S4.iT3 = mk(pw(T3, q - 2, q));
    S4.iT1q = mk(mm(pw(T1, q - 2, q), quarter, q)); S4.quarter = mk(quarter);
  } else {
    uint64_t eighth = pw((uint64_t)S % q, q - 2, q);
    for (unsigned idx = 1; idx < (unsigned)S; idx++) {
      S8.T[idx] = mk(prev(idx));
      S8.iT[idx] = mk(pw(prev(idx), q - 2, q));
    }
    S8.T[0] = S8.iT[0] = mk(0);
    S8.inv = mk(eighth);
    S8.iT1e = mk(mm(S8.iT[1].w, eighth, q));
  }
  std::vector<uint64_t> qa((size_t)S * Q), fa((size_t)S * Q);
  for (int p = 0; p < Q; p++) {  // big_pre
    uint64_t e[S >= 8 ? S : 8];
    if (!inverse) {
      for (int g = 0; g < S; g++) e[g] = in[p + (size_t)g * Q];
      if constexpr (S >= 8) {
        hx::split_fwdN<LS>(e, S8, q);
      } else {
        uint64_t o[4];
        hx::split_fwd4(e[0], e[1], e[2], e[3], S4, q, o);
        for (int g = 0; g < 4; g++) e[g] = o[g];
      }
    } else {
      for (int g = 0; g < S; g++) e[g] = in[(size_t)p * S + hx::brev_bits((unsigned)g, LS)];
    }
    for (int g = 0; g < S; g++) qa[(size_t)g * Q + p] = e[g];
  }
  for (int g = 0; g < S; g++)
    sub_transform<LOGQ>(inverse != 0, inverse ? I[g].data() : F[g].data(), q, qa.data() + (size_t)g * Q,
                        fa.data() + (size_t)g * Q);
  for (int p = 0; p < Q; p++) {  // big_post
    uint64_t e[S >= 8 ? S : 8];
    for (int g = 0; g < S; g++) e[g] = fa[(size_t)g * Q + p];
    if (!inverse) {
      for (int g = 0; g < S; g++) out[(size_t)p * S + hx::brev_bits((unsigned)g, LS)] = e[g];
    } else {
      if constexpr (S >= 8) {
        hx::split_invN<LS>(e, S8, q);
      } else {
        const uint64_t c4[4] = {e[0], e[1], e[2], e[3]};
        uint64_t a[4];
        hx::split_inv4(c4, S4, q, a);
        for (int g = 0; g < 4; g++) e[g] = a[g];
      }
      for (int g = 0; g < S; g++) out[p + (size_t)g * Q] = e[g];
    }
  }
  return 0;
}

extern "C" int big_ntt_replay(int logq, int radix, int inverse, uint64_t q, uint64_t psi, const uint64_t* in,
                              uint64_t* out)
{
  if (radix == 4 && logq == 13) return big_ntt<13, 4>(inverse, q, psi, in, out);
  if (radix == 4 && logq == 14) return big_ntt<14, 4>(inverse, q, psi, in, out);
  if (radix == 4 && logq == 15) return big_ntt<15, 4>(inverse, q, psi, in, out);
  if (radix == 8 && logq == 13) return big_ntt<13, 8>(inverse, q, psi, in, out);
  if (radix == 8 && logq == 15) return big_ntt<15, 8>(inverse, q, psi, in, out);
  if (radix == 16 && logq == 13) return big_ntt<13, 16>(inverse, q, psi, in, out);
  if (radix == 16 && logq == 15) return big_ntt<15, 16>(inverse, q, psi, in, out);
  return -1;
}

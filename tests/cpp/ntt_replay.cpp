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

template <int LOGN, int PH>
static void run_phase(bool inverse, std::vector<uint64_t>& V, std::vector<uint32_t>& NL,
                      std::vector<uint32_t>& lds, const uint64_t* in, uint64_t* out,
                      const hx::TW* tw, uint64_t q)
{
  using R = hx::RowNTT<LOGN>;
  constexpr int T = hx::Geo<LOGN>::T;
  for (unsigned tid = 0; tid < (unsigned)T; tid++) {
    uint64_t(&v)[32] = *reinterpret_cast<uint64_t(*)[32]>(&V[tid * 32]);
    uint32_t(&nl)[32] = *reinterpret_cast<uint32_t(*)[32]>(&NL[tid * 32]);
    hx::PtrIO io{in, out};
    if (inverse)
      R::template inv<PH>(tid, v, nl, lds.data(), io, tw, q);
    else
      R::template fwd<PH>(tid, v, nl, lds.data(), io, tw, q);
  }
}

template <int LOGN>
static int replay(int inverse, uint64_t q, uint64_t psi, const uint64_t* in, uint64_t* out)
{
  using G = hx::Geo<LOGN>;
  std::vector<hx::TW> f(G::TW_TOTAL), i(G::TW_TOTAL);
  uint64_t psi_inv = pw(psi, q - 2, q);
  uint64_t n_inv = pw((uint64_t)G::N % q, q - 2, q);
  hx::build_tw_tables<LOGN>(q, psi, psi_inv, n_inv, mm, f.data(), i.data());
  std::vector<uint64_t> V((size_t)G::T * 32);
  std::vector<uint32_t> NL((size_t)G::T * 32);
  std::vector<uint32_t> lds(G::LDS_WORDS, 0xdeadbeef);
  std::vector<uint64_t> inc(in, in + G::N);  // allow in == out
  const hx::TW* tw = inverse ? i.data() : f.data();
  run_phase<LOGN, 0>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 1>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 2>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 3>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 4>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 5>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 6>(inverse, V, NL, lds, inc.data(), out, tw, q);
  run_phase<LOGN, 7>(inverse, V, NL, lds, inc.data(), out, tw, q);
  return 0;
}

extern "C" int ntt_replay(int logn, int inverse, uint64_t q, uint64_t psi, const uint64_t* in,
                          uint64_t* out)
{
  switch (logn) {
    case 13: return replay<13>(inverse, q, psi, in, out);
    case 14: return replay<14>(inverse, q, psi, in, out);
    case 15: return replay<15>(inverse, q, psi, in, out);
  }
  return -1;
}

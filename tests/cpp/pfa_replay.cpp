// CPU replay of the Good-Thomas x Rader kernels' phase functions (helib_amd/csrc/pfa_core.h), thread by thread with
// a vector standing in for the LDS, so that the index arithmetic, the tables and every compile-time bound of the
// m = 21845 transform can be checked against the oracle without a GPU (built with -DHX_CHECK_BOUNDS).
// TEST INFRASTRUCTURE: built by tests/, never linked into the product library.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../helib_amd/csrc/pfa_core.h"

using namespace hx::pfa;

struct Tables {
  std::vector<uint64_t> tab;
  std::vector<uint16_t> pos2, dlog3, gpow3;
  Tables(uint64_t q, uint64_t root) : tab(TAB_WORDS), pos2(16384), dlog3(257), gpow3(256)
  {
    host::build_prime_table(q, root, tab.data());
    host::build_index_tables(pos2.data(), dlog3.data(), gpow3.data());
  }
};

template <int PH, bool INV, class Q>
static void phase(std::vector<St>& st, std::vector<uint64_t>& lds, const Args& A, const Q& q)
{
  for (unsigned tid = 0; tid < (unsigned)NT; tid++) {
    if constexpr (INV)
      inv<PH>(tid, st[tid], lds.data(), A, q);
    else
      fwd<PH>(tid, st[tid], lds.data(), A, q);
  }
}
template <int PH, class Q>
static void run_rem(std::vector<St>& st, std::vector<uint64_t>& lds, const Args& A, const Q& q)
{
  if constexpr (PH < INV_REM_PHASES) {
    for (unsigned tid = 0; tid < (unsigned)NT; tid++)
      inv_rem<PH>(tid, st[tid], lds.data(), A, q);
    run_rem<PH + 1>(st, lds, A, q);
  }
}
template <bool INV, int PH, int NPH, class Q>
static void run_all(std::vector<St>& st, std::vector<uint64_t>& lds, const Args& A, const Q& q)
{
  if constexpr (PH < NPH) {
    phase<PH, INV>(st, lds, A, q);
    run_all<INV, PH + 1, NPH>(st, lds, A, q);
  }
}

// force the generic Montgomery product (QCG) on Proth-form primes too: both arithmetics on the same rows
static int g_generic = 0;
extern "C" {
void pfa_force_generic(int on) { g_generic = on; }
int pfa_supported(uint64_t m, uint64_t q) { return host::supported(m, q) ? 1 : 0; }
// in: 16384 coefficients (canonical) -> out: 16384 evaluations in Z_m^* order
int pfa_replay_forward(uint64_t q, uint64_t root, const uint64_t* in, uint64_t* out)
{
  if (!host::supported(M, q))
    return -1;
  Tables T(q, root);
  std::vector<St> st(NT);
  std::vector<uint64_t> lds(LDS_WORDS, 0xdeadbeefdeadbeefull);
  std::vector<uint64_t> src(in, in + PHI);
  Args A{T.tab.data(), T.pos2.data(), T.dlog3.data(), T.gpow3.data(), src.data(), out};
  if (g_generic || !hx::is_proth32(q))
    run_all<false, 0, FWD_PHASES>(st, lds, A, make_qcg(q, ~(uint64_t)0 / q));
  else
    run_all<false, 0, FWD_PHASES>(st, lds, A, make_qcp(q, ~(uint64_t)0 / q));
  return 0;
}
// in: 16384 evaluations -> out: m words X[i] (before rem Phi_m and 1/m)
int pfa_replay_inverse(uint64_t q, uint64_t root, const uint64_t* in, uint64_t* out)
{
  if (!host::supported(M, q))
    return -1;
  Tables T(q, root);
  std::vector<St> st(NT);
  std::vector<uint64_t> lds(INV_LDS_WORDS, 0xdeadbeefdeadbeefull);
  std::vector<uint64_t> src(in, in + PHI);
  Args A{T.tab.data(), T.pos2.data(), T.dlog3.data(), T.gpow3.data(), src.data(), out};
  if (g_generic || !hx::is_proth32(q))
    run_all<true, 0, INV_PHASES>(st, lds, A, make_qcg(q, ~(uint64_t)0 / q));
  else
    run_all<true, 0, INV_PHASES>(st, lds, A, make_qcp(q, ~(uint64_t)0 / q));
  return 0;
}
// in: 16384 evaluations -> out: 16384 coefficients (the whole Cmodulus::iFFT: transform, rem Phi_m, 1/m)
int pfa_replay_inverse_rem(uint64_t q, uint64_t root, const uint64_t* in, uint64_t* out)
{
  if (!host::supported(M, q))
    return -1;
  Tables T(q, root);
  std::vector<St> st(NT);
  std::vector<uint64_t> lds(LDS_WORDS > REM_LDS_WORDS ? LDS_WORDS : REM_LDS_WORDS, 0xdeadbeefdeadbeefull);
  std::vector<uint64_t> src(in, in + PHI);
  Args A{T.tab.data(), T.pos2.data(), T.dlog3.data(), T.gpow3.data(), src.data(), out};
  if (g_generic || !hx::is_proth32(q))
    run_rem<0>(st, lds, A, make_qcg(q, ~(uint64_t)0 / q));
  else
    run_rem<0>(st, lds, A, make_qcp(q, ~(uint64_t)0 / q));
  return 0;
}
int pfa_table_words() { return TAB_WORDS; }
void pfa_prime_table(uint64_t q, uint64_t root, uint64_t* tab) { host::build_prime_table(q, root, tab); }
void pfa_index_tables(uint16_t* pos2, uint16_t* dlog3, uint16_t* gpow3) { host::build_index_tables(pos2, dlog3, gpow3); }
}

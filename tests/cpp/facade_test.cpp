// C++ parity test through include/helib_amd.hpp (the C++ host facade over the C ABI), checked
// bit-for-bit against the C oracle.  Built and run by tests/test_gpu_parity.py (-m gpu):
//   g++ -std=c++17 -O2 -Iinclude -Ioracle tests/cpp/facade_test.cpp -Lhelib_amd/lib -lhelib_amd
//       oracle/liboracle.so -o facade_test
// Reads like the reference's own tests/TestDoubleCRT style: build two random DoubleCRTs, apply an
// operation on both sides, compare (tests/GTestDoubleCRT.cpp in the reference tree).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>

#include "helib_amd.hpp"
extern "C" {
#include "hx_oracle.h"
}

using namespace helib_amd;
using u64 = uint64_t;

static int failures = 0;
#define EXPECT(cond, what)                                   \
  do {                                                       \
    if (!(cond)) {                                           \
      std::printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); \
      failures++;                                            \
    } else                                                   \
      std::printf("ok   %s\n", what);                        \
  } while (0)

static std::vector<u64> randomRows(std::mt19937_64& g, const Context& c, const IndexSet& s)
{
  std::vector<u64> r(s.size() * (size_t)c.getPhiM());
  for (size_t i = 0; i < s.size(); i++) {
    u64 q = (u64)c.ithPrime(s[i]);
    for (long j = 0; j < c.getPhiM(); j++)
      r[i * c.getPhiM() + j] = g() % q;
  }
  return r;
}

static int run(u64 m, int nctxt, int nspecial, int ndig)
{
  std::printf("== m=%lu ctxt=%d special=%d digits=%d\n", (unsigned long)m, nctxt, nspecial, ndig);
  Context ctx(m);
  ho_ctx* oc = ho_ctx_create(m);
  ho_primegen pg;
  ho_primegen_init(&pg, 60, (long)m);
  int L = nctxt + nspecial;
  for (int i = 0; i < L; i++) {
    long q = ho_primegen_next(&pg);
    int oi = ho_ctx_add_prime(oc, (u64)q, 0);
    long gi = ctx.addPrime((u64)q, ho_ctx_root(oc, oi));
    if (gi != oi)
      return 1;
  }
  const long N = ctx.getPhiM();
  EXPECT(N == ho_ctx_phim(oc), "phi(m)");
  IndexSet own, special, all;
  for (int i = 0; i < nctxt; i++)
    own.push_back(i);
  for (int i = nctxt; i < L; i++)
    special.push_back(i);
  all = own;
  all.insert(all.end(), special.begin(), special.end());
  std::vector<IndexSet> digits((size_t)ndig);
  for (int i = 0; i < nctxt; i++)
    digits[(size_t)(i * ndig / nctxt)].push_back(i);

  std::mt19937_64 g(m * 1315423911ull + 7);
  auto ra = randomRows(g, ctx, own), rb = randomRows(g, ctx, own);
  DoubleCRT a(ctx, own), b(ctx, own);
  a.setRows(ra);
  b.setRows(rb);

  // FFT / iFFT against Cmodulus::FFT
  std::vector<u64> ea(ra.size()), eb(rb.size()), tmp(ra.size());
  ho_dcrt_fft(oc, own.data(), nctxt, ra.data(), ea.data());
  ho_dcrt_fft(oc, own.data(), nctxt, rb.data(), eb.data());
  a.FFT();
  b.FFT();
  EXPECT(a.getRows() == ea && b.getRows() == eb, "FFT rows == oracle");
  {
    DoubleCRT t(a);
    t.iFFT();
    EXPECT(t.getRows() == ra, "iFFT(FFT(x)) == x");
  }
  // a*b + a - b, automorph
  {
    DoubleCRT t(a);
    t *= b;
    t += a;
    t -= b;
    std::vector<u64> exp(ea.size());
    for (int i = 0; i < nctxt; i++) {
      u64 q = (u64)ctx.ithPrime(i);
      u64* e = exp.data() + (size_t)i * N;
      ho_row_mul(e, ea.data() + (size_t)i * N, eb.data() + (size_t)i * N, N, q);
      ho_row_add(e, e, ea.data() + (size_t)i * N, N, q);
      ho_row_sub(e, e, eb.data() + (size_t)i * N, N, q);
    }
    EXPECT(t.getRows() == exp, "a*b + a - b");
    long k = 3;
    while (std::gcd((long)m, k) != 1)
      k += 2;
    t.automorph(k);
    std::vector<u64> au(exp.size());
    for (int i = 0; i < nctxt; i++)
      ho_row_automorph(au.data() + (size_t)i * N, exp.data() + (size_t)i * N, m, ho_ctx_zms(oc), N, (u64)k);
    EXPECT(t.getRows() == au, "automorph(k)");
    bool threw = false;
    try {
      t.automorph((m % 2 == 0) ? 2 : (long)m);
    } catch (const RuntimeError& e) {
      threw = std::strstr(e.what(), "not in Zm*") != nullptr;
    }
    EXPECT(threw, "automorph(k not in Zm*) throws RuntimeError");
  }
  // prime-set mismatch -> RuntimeError (src/DoubleCRT.cpp:243-253)
  {
    IndexSet fewer(own.begin(), own.begin() + 1);
    DoubleCRT small(ctx, fewer);
    bool threw = false;
    try {
      a += small;
    } catch (const RuntimeError&) {
      threw = true;
    }
    EXPECT(threw, "IndexSet mismatch throws RuntimeError");
  }
  // addPrimes / scaleDownToSet
  {
    DoubleCRT t(a);
    t.addPrimes(special);
    std::vector<u64> add(special.size() * (size_t)N);
    ho_dcrt_add_primes(oc, own.data(), nctxt, ea.data(), special.data(), nspecial, add.data(), nullptr);
    std::vector<u64> exp = ea;
    exp.insert(exp.end(), add.begin(), add.end());
    EXPECT(t.getRows() == exp && t.getIndexSet() == all, "addPrimes(special)");
    t.scaleDownToSet(own, 65537);
    std::vector<u64> down(ea.size());
    ho_dcrt_scale_down(oc, all.data(), L, exp.data(), special.data(), nspecial, 65537, down.data(), nullptr);
    EXPECT(t.getRows() == down && t.getIndexSet() == own, "scaleDownToSet(ctxtPrimes, p)");
  }
  // breakIntoDigits + keySwitchDigits + whole multiplyBy
  {
    std::vector<int> didx, doff;
    flatten(digits, didx, doff);
    DoubleCRT dg = a.breakIntoDigits(digits, special);
    std::vector<u64> od((size_t)ndig * L * N);
    ho_dcrt_break_into_digits(oc, own.data(), nctxt, ea.data(), didx.data(), doff.data(), ndig, all.data(), L,
                              od.data());
    EXPECT(dg.getRows() == od, "breakIntoDigits");

    std::vector<u64> kb((size_t)ndig * L * N), ka(kb.size());
    for (int d = 0; d < ndig; d++)
      for (int r = 0; r < L; r++) {
        u64 q = (u64)ctx.ithPrime(all[r]);
        for (long j = 0; j < N; j++) {
          kb[((size_t)d * L + r) * N + j] = g() % q;
          ka[((size_t)d * L + r) * N + j] = g() % q;
        }
      }
    KeySwitch W(ctx, ndig, all, kb, ka);
    auto rc1 = randomRows(g, ctx, own), rd1 = randomRows(g, ctx, own);
    DoubleCRT c1(ctx, own), d1(ctx, own);
    c1.setRows(rc1);
    d1.setRows(rd1);
    DoubleCRT o0(ctx, all), o1(ctx, all);
    multiplyBy(a, c1, b, d1, W, digits, o0, o1);
    std::vector<u64> e0((size_t)L * N), e1(e0.size());
    ho_mul_relin(oc, own.data(), nctxt, special.data(), nspecial, didx.data(), doff.data(), ndig, ea.data(),
                 rc1.data(), eb.data(), rd1.data(), kb.data(), ka.data(), e0.data(), e1.data());
    EXPECT(o0.getRows() == e0 && o1.getRows() == e1, "multiplyBy (tensorProduct + reLinearize)");
  }
  ho_ctx_destroy(oc);
  return 0;
}

int main()
{
  int dev = 0;
  if (hx_device_count(&dev) != HX_OK || dev < 1) {
    std::printf("no device: %s\n", hx_last_error());
    return 2;
  }
  if (run(4096, 4, 2, 2))
    return 1;
  if (run(1705, 3, 2, 3))
    return 1;
  std::printf("%s\n", failures ? "FAILED" : "ALL OK");
  return failures ? 1 : 0;
}

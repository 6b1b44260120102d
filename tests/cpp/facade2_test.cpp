// facade2_test.cpp -- the facade pieces a HElib caller of this path also uses: Cmodulus{FFT, iFFT}
// (include/helib/CModulus.h:137-145), the bignum DoubleCRT::toPoly (src/DoubleCRT.cpp:925-1113) and
// `namespace intel` with the HEXL shim's eight signatures (src/intelExt.h:20-59) -- a translation unit
// shaped like the reference's USE_INTEL_HEXL call sites (src/CModulus.cpp:385, 514; src/DoubleCRT.cpp:
// 144-195) compiles against helib_amd_intel.hpp unchanged.  Self-checking:
//   * Cmodulus::FFT equals the O(N^2) definition y[j] = f(zeta^{t_j}); iFFT(FFT(x)) = x; negative input
//     coefficients are reduced; the DoubleCRT transform of the same polynomial gives the same row
//   * toPoly: every BigInt coefficient reduces to the inverse-transformed row modulo each prime and lies in
//     the centred range (CRT uniqueness then makes it THE value); positive form = centred form mod Q
//   * intel::: FFTRev1(FFTFwd(x)) = x; FFTFwd + BitReverseCopy = the natural row under HEXL's root, BitReverseCopy +
//     FFTRev1 inverts it, automorph on such rows = rows of a(X^k) (the reference's call sites, src/CModulus.cpp:
//     375-426, 493-553); element-wise results against plain arithmetic, both overloads
//   usage: facade2_test m   (m a power of two or general)
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include "helib_amd.hpp"
#include "helib_amd_ctxt.hpp"
#include "helib_amd_intel.hpp"

using namespace helib_amd;

#define REQUIRE(x)                                                          \
  do {                                                                      \
    if (!(x)) {                                                             \
      printf("facade2_test FAILED at line %d: %s\n", __LINE__, #x);         \
      return 1;                                                             \
    }                                                                       \
  } while (0)

static uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((unsigned __int128)a * b % q); }
static uint64_t powmod(uint64_t a, uint64_t e, uint64_t q)
{
  uint64_t r = 1;
  for (a %= q; e; e >>= 1, a = mulmod(a, a, q))
    if (e & 1)
      r = mulmod(r, a, q);
  return r;
}

// the reference's HEXL-shaped call sites, verbatim in form (src/DoubleCRT.cpp:135-195)
struct AddFun {
  void apply(long* result, const long* a, const long* b, long n, long q) { intel::EltwiseAddMod(result, a, b, n, q); }
  void apply(long* result, const long* a, long scalar, long n, long q) { intel::EltwiseAddMod(result, a, scalar, n, q); }
};
struct SubFun {
  void apply(long* result, const long* a, const long* b, long n, long q) { intel::EltwiseSubMod(result, a, b, n, q); }
  void apply(long* result, const long* a, long scalar, long n, long q) { intel::EltwiseSubMod(result, a, scalar, n, q); }
};
struct MulFun {
  void apply(long* result, const long* a, const long* b, long n, long q) { intel::EltwiseMultMod(result, a, b, n, q); }
  void apply(long* result, const long* a, long scalar, long n, long q) { intel::EltwiseMultMod(result, a, scalar, n, q); }
};

int main(int argc, char** argv)
{
  const long m = argc > 1 ? atol(argv[1]) : 128;
  try {
    PrimeGenerator gen(60, m);
    const long q = gen.next();
    Cmodulus cm((unsigned long)m, q);
    const size_t n = cm.getPhiM();
    REQUIRE(cm.getQ() == q && cm.getM() == (unsigned long)m && cm.getRoot() > 1);
    // Z_m^* in increasing order and zeta: root for a power of two (evaluation at root^(2j+1)), root^2 otherwise
    const bool pow2 = (m & (m - 1)) == 0;
    std::vector<long> zms;
    for (long t = 1; t < m; t++)
      if (std::gcd(t, m) == 1)
        zms.push_back(t);
    REQUIRE(zms.size() == n);
    const uint64_t root = (uint64_t)cm.getRoot();
    const uint64_t zeta = pow2 ? root : mulmod(root, root, (uint64_t)q);
    uint64_t seed = 99;
    auto next = [&]() {
      seed = seed * 6364136223846793005ull + 1442695040888963407ull;
      return seed >> 11;
    };
    std::vector<long> x(n), y, back;
    for (auto& v : x)
      v = (long)(next() % (uint64_t)q) - q / 2;    // signed coefficients: FFT reduces them
    cm.FFT(y, x);
    REQUIRE(y.size() == n);
    const size_t ncheck = n <= 256 ? n : 24;
    for (size_t c = 0; c < ncheck; c++) {
      const size_t j = n <= 256 ? c : (size_t)(next() % n);
      const uint64_t pt = powmod(zeta, (uint64_t)zms[j], (uint64_t)q);
      uint64_t acc = 0;
      for (size_t i = n; i-- > 0;) {
        long r = x[i] % q;
        acc = (mulmod(acc, pt, (uint64_t)q) + (uint64_t)(r < 0 ? r + q : r)) % (uint64_t)q;
      }
      REQUIRE((uint64_t)y[j] == acc);
    }
    cm.iFFT(back, y);
    for (size_t i = 0; i < n; i++) {
      long r = x[i] % q;
      REQUIRE(back[i] == (r < 0 ? r + q : r));
    }
    bool threw = false;
    try {
      std::vector<long> bad(n, q);
      cm.iFFT(back, bad);
    } catch (const InvalidArgument&) {
      threw = true;
    }
    REQUIRE(threw);

    // ---- bignum toPoly on a 3-prime, batch-2 DoubleCRT ----
    Context ctx((uint64_t)m);
    PrimeGenerator g2(60, m);
    std::vector<uint64_t> qs;
    IndexSet all;
    for (int i = 0; i < 3; i++) {
      qs.push_back((uint64_t)g2.next());
      all.push_back((int)ctx.addPrime(qs.back()));
    }
    REQUIRE((long)qs[0] == q);
    const int B = 2;
    DoubleCRT d(ctx, all, B);
    std::vector<uint64_t> rows(3 * (size_t)B * n);
    for (size_t r = 0; r < 3; r++)
      for (size_t i = 0; i < (size_t)B * n; i++)
        rows[r * B * n + i] = next() % qs[r];
    d.setRows(rows);
    DoubleCRT coef(d);
    coef.iFFT();
    const std::vector<uint64_t> crow = coef.getRows();
    BigInt Q(1);
    for (uint64_t qq : qs)
      Q.mulAdd(qq, 0);
    const BigInt half = Q.halfUp();
    for (int b = 0; b < B; b++) {
      const std::vector<BigInt> poly = d.toPoly(nullptr, false, b), pos = d.toPoly(nullptr, true, b);
      REQUIRE(poly.size() == n && pos.size() == n);
      size_t negs = 0;
      for (size_t j = 0; j < n; j++) {
        for (size_t r = 0; r < 3; r++) {
          REQUIRE(poly[j].mod(qs[r]) == crow[(r * B + (size_t)b) * n + j]);
          REQUIRE(pos[j].mod(qs[r]) == crow[(r * B + (size_t)b) * n + j]);
        }
        REQUIRE(!pos[j].negative() && BigInt::cmpMag(pos[j], Q) < 0);
        // centred: -(Q - half) .. half - 1, i.e. |v| < half for v >= 0 and |v| <= Q - half for v < 0
        if (poly[j].negative()) {
          negs++;
          REQUIRE(BigInt::cmpMag(poly[j], BigInt::subMag(Q, half)) <= 0);
          REQUIRE(BigInt::cmpMag(BigInt::subMag(Q, poly[j]), pos[j]) == 0);
        } else {
          REQUIRE(BigInt::cmpMag(poly[j], half) < 0 && BigInt::cmpMag(poly[j], pos[j]) == 0);
        }
      }
      REQUIRE(negs > n / 4 && negs < 3 * n / 4);   // uniform rows: about half of the lifts are negative
    }
    // restricted to a subset of the primes (toPoly(poly, s)), and decimal output
    IndexSet two{all[0], all[2]};
    const std::vector<BigInt> sub = d.toPoly(&two, false, 1);
    for (size_t j = 0; j < n; j += 7) {
      REQUIRE(sub[j].mod(qs[0]) == crow[(0 * B + 1) * n + j] && sub[j].mod(qs[2]) == crow[(2 * B + 1) * n + j]);
      const std::string dec = sub[j].toString();
      REQUIRE(!dec.empty() && (dec == "0" || dec[0] == '-' || (dec[0] >= '1' && dec[0] <= '9')));
    }
    BigInt t(123456789012345678ull);
    t.mulAdd(1000000000000000000ull, 987654321098765432ull);
    REQUIRE(t.toString() == "123456789012345678987654321098765432" && t.negated().toString() == "-123456789012345678987654321098765432");
    // the first row of the DoubleCRT transform is the Cmodulus transform of the same coefficients
    {
      std::vector<long> c0(n), e0;
      for (size_t j = 0; j < n; j++)
        c0[j] = (long)crow[(0 * B + 0) * n + j];
      cm.FFT(e0, c0);
      for (size_t j = 0; j < n; j++)
        REQUIRE((uint64_t)e0[j] == rows[(0 * B + 0) * n + j]);
    }

    // ---- namespace intel (power-of-two n only, as HEXL) ----
    if (pow2) {
      const long nn = (long)n;
      std::vector<long> a(n), b2(n), r1(n), r2(n);
      for (size_t i = 0; i < n; i++) {
        a[i] = (long)(next() % (uint64_t)q);
        b2[i] = (long)(next() % (uint64_t)q);
      }
      intel::FFTFwd(r1.data(), a.data(), nn, q);
      intel::FFTRev1(r2.data(), r1.data(), nn, q);
      REQUIRE(r2 == a);
      // The call sites, not just the round trip.  Cmodulus::FFT_aux's HEXL branch (src/CModulus.cpp:375-385)
      // is followed by BitReverseCopy (:421-426); Cmodulus::iFFT bit-reverses BEFORE intel::FFTRev1
      // (:510-514): HEXL's forward transform delivers, and its inverse consumes, bit-reversed order.
      // So FFTFwd + BitReverseCopy must be the NATURAL row y[j] = f(psi^(2j+1)) that DoubleCRT::automorph
      // (src/DoubleCRT.cpp:1160-1202) indexes, under HEXL's root (the smallest primitive 2n-th root).
      int bits = 0;
      while ((1L << bits) < nn)
        bits++;
      auto brc = [&](const std::vector<long>& A) {   // B[rev(i)] = A[i]  (src/CModulus.cpp:284-299)
        std::vector<long> Bv(A.size());
        for (size_t i = 0; i < A.size(); i++) {
          size_t r = 0;
          for (int t = 0; t < bits; t++)
            r |= ((i >> t) & 1u) << (bits - 1 - t);
          Bv[r] = A[i];
        }
        return Bv;
      };
      uint64_t psi = root, cur = root;  // walk every primitive 2n-th root = the odd powers of any one
      const uint64_t rsq = mulmod(root, root, (uint64_t)q);
      for (long i = 0; i < nn; i++, cur = mulmod(cur, rsq, (uint64_t)q))
        psi = cur < psi ? cur : psi;
      const std::vector<long> nat = brc(r1);
      Cmodulus hm((unsigned long)m, q, (long)psi);
      std::vector<long> ynat;
      hm.FFT(ynat, a);
      REQUIRE(ynat == nat);                           // shim forward + BitReverseCopy = the engine's natural row
      for (size_t c = 0; c < 8; c++) {                // ... which is f(psi^(2j+1)) by definition
        const size_t j = (size_t)(next() % n);
        const uint64_t pt = powmod(psi, 2 * j + 1, (uint64_t)q);
        uint64_t acc = 0;
        for (size_t i = n; i-- > 0;)
          acc = (mulmod(acc, pt, (uint64_t)q) + (uint64_t)a[i]) % (uint64_t)q;
        REQUIRE((uint64_t)nat[j] == acc);
      }
      std::vector<long> rev_in = brc(nat), back2(n);  // iFFT's order: BitReverseCopy, then FFTRev1
      intel::FFTRev1(back2.data(), rev_in.data(), nn, q);
      REQUIRE(back2 == a);
      // automorph on rows that came through the shim path: X -> X^k on the coefficients (negacyclic), rows of
      // both through FFTFwd + BitReverseCopy, against DoubleCRT::automorph of the first
      {
        const long k = 5 % (2 * nn) == 1 ? 3 : 5;
        std::vector<long> ak(n, 0), rk(n);
        for (long i = 0; i < nn; i++) {
          const long e = (i * k) % (2 * nn);
          if (e < nn)
            ak[(size_t)e] = a[(size_t)i];
          else
            ak[(size_t)(e - nn)] = a[(size_t)i] ? q - a[(size_t)i] : 0;
        }
        intel::FFTFwd(rk.data(), ak.data(), nn, q);
        const std::vector<long> natk = brc(rk);
        Context hc((uint64_t)m);
        IndexSet one{(int)hc.addPrime((uint64_t)q, psi)};
        DoubleCRT dr(hc, one, 1);
        std::vector<uint64_t> rowsv(n);
        for (size_t j = 0; j < n; j++)
          rowsv[j] = (uint64_t)nat[j];
        dr.setRows(rowsv);
        dr.automorph(k);
        const std::vector<uint64_t> got = dr.getRows();
        for (size_t j = 0; j < n; j++)
          REQUIRE(got[j] == (uint64_t)natk[j]);
      }
      AddFun add;
      SubFun sub2;
      MulFun mul;
      const long sc = 123456789 % q;
      add.apply(r1.data(), a.data(), b2.data(), nn, q);
      for (size_t i = 0; i < n; i++)
        REQUIRE(r1[i] == (long)(((uint64_t)a[i] + (uint64_t)b2[i]) % (uint64_t)q));
      add.apply(r1.data(), a.data(), sc, nn, q);
      for (size_t i = 0; i < n; i++)
        REQUIRE(r1[i] == (long)(((uint64_t)a[i] + (uint64_t)sc) % (uint64_t)q));
      sub2.apply(r1.data(), a.data(), b2.data(), nn, q);
      for (size_t i = 0; i < n; i++)
        REQUIRE(r1[i] == (long)(((uint64_t)a[i] + (uint64_t)q - (uint64_t)b2[i]) % (uint64_t)q));
      sub2.apply(r1.data(), a.data(), sc, nn, q);
      for (size_t i = 0; i < n; i++)
        REQUIRE(r1[i] == (long)(((uint64_t)a[i] + (uint64_t)q - (uint64_t)sc) % (uint64_t)q));
      mul.apply(r1.data(), a.data(), b2.data(), nn, q);
      for (size_t i = 0; i < n; i++)
        REQUIRE(r1[i] == (long)mulmod((uint64_t)a[i], (uint64_t)b2[i], (uint64_t)q));
      mul.apply(a.data(), a.data(), sc, nn, q);   // in place, as do_mul's call site allows
      for (size_t i = 0; i < n; i++)
        REQUIRE(a[i] == (long)mulmod((uint64_t)r2[i], (uint64_t)sc, (uint64_t)q));
    }
    printf("facade2_test OK (m=%ld)\n", m);
    return 0;
  } catch (const std::exception& e) {
    printf("facade2_test exception: %s\n", e.what());
    return 1;
  }
}

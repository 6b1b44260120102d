// GPU check of include/helib_amd_keys.hpp: key generation, PubKey::Encrypt, Ctxt::multiplyBy,
// smartAutomorph (one step and two steps along the key-switch map) and SecKey::Decrypt, all
// driven from C++ over the C ABI, against plain polynomial arithmetic modulo (X^N + 1, p).
//   keys_test <m> <p> <bits> <measure>
#include <cstdio>
#include <cstdlib>

#include "helib_amd_keys.hpp"

using namespace helib_amd;

static std::vector<long> negacyclic(const std::vector<long>& a, const std::vector<long>& b, long p)
{
  size_t n = a.size();
  std::vector<long> out(n, 0);
  for (size_t i = 0; i < n; i++) {
    if (a[i] == 0)
      continue;
    for (size_t j = 0; j < n; j++) {
      long t = (long)((unsigned __int128)a[i] * (unsigned long)b[j] % (unsigned long)p);
      size_t k = i + j;
      if (k < n)
        out[k] = (out[k] + t) % p;
      else
        out[k - n] = (out[k - n] + p - t) % p;
    }
  }
  return out;
}
// f(X) -> f(X^k) modulo X^N + 1
static std::vector<long> automorph(const std::vector<long>& a, long k, long p)
{
  size_t n = a.size();
  std::vector<long> out(n, 0);
  for (size_t i = 0; i < n; i++) {
    size_t e = (size_t)((unsigned __int128)i * (unsigned long)k % (2 * n));
    if (e < n)
      out[e] = (out[e] + a[i]) % p;
    else
      out[e - n] = (out[e - n] + p - a[i]) % p;
  }
  return out;
}
static std::vector<double> negacyclic_d(const std::vector<double>& a, const std::vector<double>& b)
{
  size_t n = a.size();
  std::vector<double> out(n, 0.0);
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < n; j++) {
      size_t k = i + j;
      if (k < n)
        out[k] += a[i] * b[j];
      else
        out[k - n] -= a[i] * b[j];
    }
  return out;
}
static double maxdiff(const std::vector<double>& a, const std::vector<double>& b)
{
  double d = 0;
  for (size_t i = 0; i < a.size(); i++)
    d = std::max(d, std::fabs(a[i] - b[i]));
  return d;
}
static int ckks_main(long m, long bits, bool measure);
#define REQUIRE(c)                                             \
  do {                                                         \
    if (!(c)) {                                                \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                \
    }                                                          \
  } while (0)

int main(int argc, char** argv)
{
  if (argc < 5)
    return 2;
  Ctxt::deferNorms() = getenv("HX_TEST_DEFER_NORMS") != nullptr;   // measured norms read back lazily (LazyLn)
  long m = atol(argv[1]), p = atol(argv[2]), bits = atol(argv[3]);
  bool measure = atol(argv[4]) != 0;
  if (p == -1)
    return ckks_main(m, bits, measure);
  try {
    ChainContext cc(m, p, 1, bits, 3);
    auto dev = cc.makeDeviceContext(0);
    SecKey sk(cc, *dev, 12345);
    sk.GenSecKey(2);
    sk.GenKeySWmatrix(1, 3);
    sk.setKeySwitchMap();
    REQUIRE(sk.keySwitching.size() == 2 && sk.keys.relin && sk.keys.automorph.count(3));
    REQUIRE(sk.keys.isReachable(9) && sk.keys.firstStep(9) == 3 && !sk.keys.isReachable(m - 1));
    long nz = 0;
    for (long v : sk.sKey) {
      REQUIRE(v >= -1 && v <= 1);
      nz += v != 0;
    }
    REQUIRE(nz > cc.phim / 4 && nz < 3 * cc.phim / 4);
    REQUIRE(sk.sampler.embeddingLargestCoeff(sk.sKey) <= sk.skBound);

    std::mt19937_64 rng(7);
    std::vector<long> ma((size_t)cc.phim), mb((size_t)cc.phim);
    for (auto& v : ma)
      v = (long)(rng() % (uint64_t)p);
    for (auto& v : mb)
      v = (long)(rng() % (uint64_t)p);
    Ctxt ca = sk.Encrypt(ma), cb = sk.Encrypt(mb);
    ca.measure = cb.measure = measure;
    REQUIRE(sk.Decrypt(ca) == ma && sk.Decrypt(cb) == mb);
    REQUIRE(ca.parts.size() == 2 && ca.primeSet.size() == cc.ctxtPrimes.size());

    ca.multiplyBy(cb);
    std::vector<long> prod = negacyclic(ma, mb, p);
    REQUIRE(ca.parts.size() == 2);
    REQUIRE(sk.Decrypt(ca) == prod);
    {  // before relinearisation the s^2 part decrypts with s^2
      Ctxt c1 = sk.Encrypt(ma), c2 = sk.Encrypt(mb);
      c1.measure = c2.measure = measure;
      c1.multLowLvl(c2);
      REQUIRE(c1.parts.size() == 3);
      REQUIRE(sk.Decrypt(c1) == prod);
    }
    Ctxt sum = ca;
    sum.addCtxt(ca);
    std::vector<long> twice(prod);
    for (auto& v : twice)
      v = 2 * v % p;
    REQUIRE(sk.Decrypt(sum) == twice);

    ca.smartAutomorph(3);
    std::vector<long> rot = automorph(prod, 3, p);
    REQUIRE(sk.Decrypt(ca) == rot);
    ca.smartAutomorph(9);   // two steps of 3 along the map
    REQUIRE(sk.Decrypt(ca) == automorph(rot, 9, p));
    REQUIRE(std::isfinite((double)ca.lnNoise) && ca.lnNoise > 0);

    // ---- batched encryption / decryption: element b of EncryptBatch(B) is the b-th of B consecutive Encrypt() calls
    // (same key, same sampler state: two SecKey objects built from one seed), DecryptBatch returns every element,
    // and a batched product decrypts to the B plaintext products
    {
      const int B = 3;
      const size_t n = (size_t)cc.phim;
      SecKey s1(cc, *dev, 4242), s2(cc, *dev, 4242);
      s1.GenSecKey(2);
      s2.GenSecKey(2);
      std::vector<long> msgs((size_t)B * n), other((size_t)B * n);
      for (auto& v : msgs)
        v = (long)(rng() % (uint64_t)p);
      for (auto& v : other)
        v = (long)(rng() % (uint64_t)p);
      Ctxt batch = s2.EncryptBatch(msgs, B);
      REQUIRE(batch.parts.size() == 2 && batch.parts.begin()->second.batch() == B);
      double worst = -1e300;
      for (int b = 0; b < B; b++) {
        Ctxt one = s1.Encrypt(std::vector<long>(msgs.begin() + (size_t)b * n, msgs.begin() + (size_t)(b + 1) * n));
        worst = std::max(worst, (double)one.lnNoise);
        for (auto& kv : one.parts) {
          const std::vector<uint64_t> r1 = kv.second.getRows(), rB = batch.parts.at(kv.first).getRows();   // [row][B][n]
          const size_t rows = r1.size() / n;
          for (size_t r = 0; r < rows; r++)
            for (size_t j = 0; j < n; j++)
              REQUIRE(rB[(r * (size_t)B + (size_t)b) * n + j] == r1[r * n + j]);
        }
      }
      REQUIRE(std::fabs((double)batch.lnNoise - worst) < 1e-9);   // the batch's estimate = its largest element's
      REQUIRE(s2.DecryptBatch(batch) == msgs);
      Ctxt ob = s2.EncryptBatch(other, B);
      batch.measure = ob.measure = measure;
      batch.multiplyBy(ob);
      const std::vector<long> got = s2.DecryptBatch(batch);
      for (int b = 0; b < B; b++) {
        const std::vector<long> want = negacyclic(std::vector<long>(msgs.begin() + (size_t)b * n, msgs.begin() + (size_t)(b + 1) * n),
                                                  std::vector<long>(other.begin() + (size_t)b * n, other.begin() + (size_t)(b + 1) * n), p);
        REQUIRE(std::equal(want.begin(), want.end(), got.begin() + (size_t)b * n));
      }
      bool threw = false;
      try {
        s2.EncryptBatch(std::vector<long>(n + 1), 1);
      } catch (const InvalidArgument&) {
        threw = true;
      }
      REQUIRE(threw);
      // key material WITHOUT the secret key: the holder encrypts and multiplies, cannot decrypt; the owner of the
      // full key decrypts what it made
      const std::vector<uint64_t> pub = s2.exportKeys(false), full = s2.exportKeys();
      REQUIRE(pub.size() == full.size() && pub != full);
      SecKey only(cc, *dev, 99);
      only.importKeys(pub.data(), pub.size());
      REQUIRE(only.sKey.empty() && only.keys.relin);
      const std::vector<long> m0(msgs.begin(), msgs.begin() + n), m1(other.begin(), other.begin() + n);
      Ctxt x = only.Encrypt(m0), y = only.Encrypt(m1);
      x.measure = y.measure = measure;
      x.multiplyBy(y);
      threw = false;
      try {
        only.Decrypt(x);
      } catch (const LogicError&) {
        threw = true;
      }
      REQUIRE(threw);
      REQUIRE(s2.Decrypt(x) == negacyclic(m0, m1, p));
      // ... and it cannot make keys either: a key-switching matrix "under s = 0" or a fresh secret key
      // next to the imported matrices would give wrong ciphertexts with no error
      for (int what = 0; what < 3; what++) {
        threw = false;
        try {
          if (what == 0)
            only.GenKeySWmatrix(1, 3);       // (what add1DMatrices and its siblings call per family member)
          else if (what == 1)
            only.GenSecKey(2);
          else
            only.GenKeySWmatrix(3, 1);
        } catch (const LogicError&) {
          threw = true;
        }
        REQUIRE(threw);
      }
      REQUIRE(only.sKey.empty() && !only.haveKeySWmatrix(1, 3));
      // the has-secret word of the header and the polynomial must agree in every coefficient
      std::vector<uint64_t> bad = pub;
      bad[SecKey::KEYS_HEADER + 5] = 1;      // one coefficient present in a "public" blob
      SecKey other2(cc, *dev, 98);
      threw = false;
      try {
        other2.importKeys(bad.data(), bad.size());
      } catch (const InvalidArgument&) {
        threw = true;
      }
      REQUIRE(threw && !other2.pubEncrKey0);
      bad = full;
      bad[8] = 0;                            // "no secret" over a blob that carries one
      threw = false;
      try {
        other2.importKeys(bad.data(), bad.size());
      } catch (const InvalidArgument&) {
        threw = true;
      }
      REQUIRE(threw && !other2.pubEncrKey0);
      other2.importKeys(full.data(), full.size());
      REQUIRE(!other2.sKey.empty() && other2.Decrypt(x) == negacyclic(m0, m1, p));
    }
    dev->sync();
  } catch (const std::exception& ex) {
    fprintf(stderr, "exception: %s\n", ex.what());
    return 1;
  }
  printf("keys_test OK\n");
  return 0;
}

// CKKS: ContextBuilder<CKKS>().m(m).precision(20).bits(bits).c(3); CKKSencrypt of scaled real
// polynomials, products over two levels, a sum across different scaling factors and a difference,
// decoded by DecryptCKKS and compared with double arithmetic on the encoded values; every error
// must also stay below the bound the ciphertext reports (noiseBound / ratFactor).
static int ckks_main(long m, long bits, bool measure)
{
  try {
    const long precision = 20;
    ChainContext cc(m, -1, precision, bits, 3, 3.2, 10.0, 0, 3, 0, true);
    REQUIRE(cc.ckks && cc.ptxtSpace == 1 && cc.p == -1);
    auto dev = cc.makeDeviceContext(0);
    SecKey sk(cc, *dev, 777);
    sk.GenSecKey(2);
    const size_t n = (size_t)cc.phim;
    const double f = std::ldexp(1.0, (int)precision);
    std::mt19937_64 rng(3);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    auto draw = [&](std::vector<long>& scaled, std::vector<double>& enc) {
      scaled.resize(n);
      enc.resize(n);
      for (size_t i = 0; i < n; i++) {
        scaled[i] = std::lround(U(rng) / (double)n * f);
        enc[i] = (double)scaled[i] / f;        // the value actually encrypted
      }
    };
    std::vector<long> pa, pb, pc;
    std::vector<double> a, b, c;
    draw(pa, a), draw(pb, b), draw(pc, c);
    Ctxt ca = sk.CKKSencrypt(pa, 1.0, f), cb = sk.CKKSencrypt(pb, 1.0, f), c3 = sk.CKKSencrypt(pc, 1.0, f);
    ca.measure = cb.measure = c3.measure = measure;
    REQUIRE(ca.ptxtSpace == 1 && ca.ptxtMag == 1.0);
    REQUIRE(ca.lnRatFactor >= ca.lnNoise + precision * std::log(2.0) - 1e-9);
    auto within = [&](const Ctxt& ct, const std::vector<double>& want) {
      double err = maxdiff(sk.DecryptCKKS(ct), want), bound = std::exp(ct.lnNoise - ct.lnRatFactor);
      if (!(err <= bound))
        fprintf(stderr, "error %g above the reported bound %g\n", err, bound);
      return err <= bound;
    };
    REQUIRE(within(ca, a));
    REQUIRE(maxdiff(sk.DecryptCKKS(ca), a) < std::ldexp(1.0, -(int)precision));

    ca.multiplyBy(cb);                              // level 1: no mod-switch for fresh operands
    std::vector<double> ab = negacyclic_d(a, b);
    REQUIRE(ca.parts.size() == 2 && within(ca, ab));
    Ctxt sum = ca, diff = ca;
    sum.addCtxt(c3);                                // factors f^2-ish and f: equalizeRationalFactors
    REQUIRE(std::fabs(sum.lnRatFactor - ca.lnRatFactor) < 40.0 && sum.ptxtMag == 2.0);
    std::vector<double> want(n);
    for (size_t i = 0; i < n; i++)
      want[i] = ab[i] + c[i];
    REQUIRE(within(sum, want));
    diff.addCtxt(c3, true);
    for (size_t i = 0; i < n; i++)
      want[i] = ab[i] - c[i];
    REQUIRE(within(diff, want));
    Ctxt sq = ca;
    sq.multiplyBy(ca);                              // level 2: both operands mod-switched first
    REQUIRE(within(sq, negacyclic_d(ab, ab)));
    size_t kept = 0;
    for (int i : sq.primeSet)
      kept += std::find(cc.ctxtPrimes.begin(), cc.ctxtPrimes.end(), i) != cc.ctxtPrimes.end();
    REQUIRE(kept <= cc.ctxtPrimes.size());
    dev->sync();
  } catch (const std::exception& ex) {
    fprintf(stderr, "exception: %s\n", ex.what());
    return 1;
  }
  printf("keys_test OK\n");
  return 0;
}

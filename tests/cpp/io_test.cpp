// Ctxt::writeTo / Ctxt::read from C++ (include/helib_amd_io.hpp): ciphertexts of the C++ host cross the
// reference's binary layout and come back as working ciphertexts -- same prime set, factors and noise
// estimate, decrypting to the same plaintext, and usable in further multiplications; malformed blobs raise
// IOError.  Every blob, followed by the JSON text of the same ciphertext, is also appended to <out> as
// [int64 length][bytes] for the python side, which parses the blob with helib_amd.wire (pinned on the reference's
// own fixture), writes it back byte for byte and compares the JSON with its own.
//   io_test <m> <p> <bits> <out>        p = -1: CKKS
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "helib_amd_io.hpp"
#include "helib_amd_keys.hpp"

using namespace helib_amd;
typedef std::vector<long> Poly;

static Poly negacyclic(const Poly& a, const Poly& b, long p)
{
  size_t n = a.size();
  Poly out(n, 0);
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < n; j++) {
      long t = (long)((unsigned __int128)a[i] * (unsigned long)b[j] % (unsigned long)p);
      size_t k = i + j;
      if (k < n)
        out[k] = (out[k] + t) % p;
      else
        out[k - n] = (out[k - n] + p - t) % p;
    }
  return out;
}
#define REQUIRE(c)                                                   \
  do {                                                               \
    if (!(c)) {                                                      \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                      \
    }                                                                \
  } while (0)
static void dump(std::ofstream& f, const std::string& blob)
{
  int64_t n = (int64_t)blob.size();
  f.write(reinterpret_cast<const char*>(&n), 8);
  f.write(blob.data(), n);
}
template <class F>
static bool raisesIOError(F f)
{
  try {
    f();
  } catch (const wire::IOError&) {
    return true;
  }
  return false;
}

int main(int argc, char** argv)
{
  if (argc < 5)
    return 2;
  Ctxt::deferNorms() = getenv("HX_TEST_DEFER_NORMS") != nullptr;   // measured norms read back lazily (LazyLn)
  const long m = atol(argv[1]), p = atol(argv[2]), bits = atol(argv[3]);
  std::ofstream out(argv[4], std::ios::binary);
  try {
    if (p == -1) {
      const long precision = 20;
      ChainContext cc(m, -1, precision, bits, 3, 3.2, 10.0, 0, 3, 0, true);
      auto dev = cc.makeDeviceContext(0);
      SecKey sk(cc, *dev, 31);
      sk.GenSecKey(2);
      const size_t n = (size_t)cc.phim;
      const double f = std::ldexp(1.0, (int)precision);
      Poly pa(n), pb(n);
      std::vector<double> a(n), b(n);
      std::mt19937_64 rng(9);
      for (size_t i = 0; i < n; i++) {
        pa[i] = (long)(rng() % 2001) - 1000, pb[i] = (long)(rng() % 2001) - 1000;
        a[i] = (double)pa[i] / f, b[i] = (double)pb[i] / f;
      }
      Ctxt ca = sk.CKKSencrypt(pa, 1.0, f), cb = sk.CKKSencrypt(pb, 1.0, f);
      ca.multiplyBy(cb);
      std::string blob = writeTo(ca);
      dump(out, blob);
      std::string js = writeToJSON(ca);
      dump(out, js);
      REQUIRE(writeTo(readCtxtFromJSON(js, cc, *dev, sk.keys)) == blob);
      Ctxt rb = readCtxtFrom(blob.data(), blob.size(), cc, *dev, sk.keys);
      REQUIRE(rb.primeSet == ca.primeSet && rb.ptxtMag == ca.ptxtMag);
      REQUIRE(std::fabs(rb.lnRatFactor - ca.lnRatFactor) < 1e-9 && std::fabs(rb.lnNoise - ca.lnNoise) < 1e-9);
      std::vector<double> d1 = sk.DecryptCKKS(ca), d2 = sk.DecryptCKKS(rb);
      for (size_t i = 0; i < n; i++)
        REQUIRE(std::fabs(d1[i] - d2[i]) <= 1e-12 * (1.0 + std::fabs(d1[i])));
      rb.multiplyBy(cb);                    // a level further down with the restored bookkeeping
      std::vector<double> d3 = sk.DecryptCKKS(rb);
      double err = 0, bound = std::exp(rb.lnNoise - rb.lnRatFactor);
      std::vector<double> ab(n, 0.0), abb(n, 0.0);
      for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) {
          size_t k = i + j;
          if (k < n)
            ab[k] += a[i] * b[j];
          else
            ab[k - n] -= a[i] * b[j];
        }
      for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) {
          size_t k = i + j;
          if (k < n)
            abb[k] += ab[i] * b[j];
          else
            abb[k - n] -= ab[i] * b[j];
        }
      for (size_t i = 0; i < n; i++)
        err = std::max(err, std::fabs(d3[i] - abb[i]));
      REQUIRE(err <= bound);
      printf("io_test OK\n");
      return 0;
    }
    ChainContext cc(m, p, 1, bits, 3);
    auto dev = cc.makeDeviceContext(0);
    SecKey sk(cc, *dev, 77);
    sk.GenSecKey(3);                        // (the restored 3-part product is multiplied again: parts up to s^3)
    const size_t n = (size_t)cc.phim;
    std::mt19937_64 rng(3);
    Poly a(n), b(n);
    for (auto& v : a)
      v = (long)(rng() % (uint64_t)p);
    for (auto& v : b)
      v = (long)(rng() % (uint64_t)p);
    Ctxt ca = sk.Encrypt(a), cb = sk.Encrypt(b);
    const Poly ab = negacyclic(a, b, p);

    // a fresh ciphertext, a relinearised product (on ctxt + special primes) and a 3-part product
    Ctxt prod = ca;
    prod.multiplyBy(cb);
    Ctxt low = ca;
    low.multLowLvl(cb);
    const Ctxt* cases[3] = {&ca, &prod, &low};
    const Poly* wants[3] = {&a, &ab, &ab};
    for (int i = 0; i < 3; i++) {
      const Ctxt& c = *cases[i];
      std::string blob = writeTo(c);
      dump(out, blob);
      size_t used = 0;
      Ctxt r = readCtxtFrom(blob.data(), blob.size(), cc, *dev, sk.keys, &used);
      REQUIRE(used == blob.size());
      REQUIRE(r.primeSet == c.primeSet && r.ptxtSpace == c.ptxtSpace && r.intFactor == c.intFactor);
      REQUIRE(r.parts.size() == c.parts.size() && std::fabs(r.lnNoise - c.lnNoise) < 1e-9);
      for (auto& kv : c.parts) {             // the same rows prime by prime (the restored object lists them ascending)
        REQUIRE(r.parts.count(kv.first));
        wire::Rows x = wire::fromPoly(r.parts.at(kv.first), n), y = wire::fromPoly(kv.second, n);
        REQUIRE(x.idx == y.idx && x.data == y.data);
      }
      REQUIRE(sk.Decrypt(r) == *wants[i]);
      REQUIRE(writeTo(r) == blob);          // and the same bytes again
      std::string js = writeToJSON(c);      // the JSON form: same object
      dump(out, js);
      Ctxt rj = readCtxtFromJSON(js, cc, *dev, sk.keys);
      REQUIRE(writeTo(rj) == blob && writeToJSON(rj) == js && sk.Decrypt(rj) == *wants[i]);
      Ctxt next = r;                        // the restored object keeps working
      next.multiplyBy(cb);
      REQUIRE(sk.Decrypt(next) == negacyclic(*wants[i], b, p));
    }
    // malformed blobs
    std::string blob = writeTo(prod);
    REQUIRE(raisesIOError([&] { readCtxtFrom(blob.data(), blob.size() - 5, cc, *dev, sk.keys); }));
    {
      std::string bad = blob;
      bad[3] = 'X';                         // header eye catcher
      REQUIRE(raisesIOError([&] { readCtxtFrom(bad.data(), bad.size(), cc, *dev, sk.keys); }));
    }
    {
      wire::CtxtDesc d = wire::describe(prod);
      d.parts[0].rows.data[5] = cc.primes[(size_t)d.parts[0].rows.idx[0]];   // a residue equal to its prime
      REQUIRE(raisesIOError([&] { wire::restore(d, cc, *dev, sk.keys); }));
      d = wire::describe(prod);
      d.primeSet.pop_back();                // a part on more primes than the ciphertext says
      REQUIRE(raisesIOError([&] { wire::restore(d, cc, *dev, sk.keys); }));
      d = wire::describe(prod);
      d.primeSet.back() = (long)cc.primes.size();   // a prime the context does not have
      REQUIRE(raisesIOError([&] { wire::restore(d, cc, *dev, sk.keys); }));
    }
    // xdouble conversions
    for (double ln : {-700.0, -1.0, 0.0, 39.5, 78.0, 79.1, 400.0, 5000.0}) {
      wire::XDouble x = wire::xdFromLn(ln);
      REQUIRE(std::fabs(x.mantissa) >= std::ldexp(1.0, -57) && std::fabs(x.mantissa) < std::ldexp(1.0, 57));
      REQUIRE(std::fabs(wire::lnOf(x) - ln) < 1e-9 * (1.0 + std::fabs(ln)));
    }
    REQUIRE(wire::xdOf(0.0).mantissa == 0.0 && wire::valueOf(wire::xdOf(1e40)) == 1e40 && wire::xdOf(1e40).exponent == 1);
  } catch (const std::exception& ex) {
    fprintf(stderr, "exception: %s\n", ex.what());
    return 1;
  }
  printf("io_test OK\n");
  return 0;
}

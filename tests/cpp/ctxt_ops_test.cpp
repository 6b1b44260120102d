// The wider Ctxt operations of include/helib_amd_ctxt.hpp driven from C++ over the C ABI and checked against
// plain polynomial arithmetic modulo (Phi_m(X), p), any m: multiplyBy2 / cube / power (parts up to s^3, one
// relinearisation through keySwitchPart), totalProduct / incrementalProduct / innerProduct, hoisted
// rotations (BasicAutomorphPrecon, one and two steps along the key-switch map), frobeniusAutomorph,
// multByConstant / addConstant with scalars and DoubleCRT constants, capacity / isCorrect; and the CKKS
// forms (multByConstantCKKS, addConstantCKKS, complex conjugation).
//   ctxt_ops_test <m> <p> <bits> <measure>        p = -1: CKKS
// Links against libhelib_amd.so on a GPU box (-m gpu suite) or against the CPU mock of the C ABI
// (tests/cpp/hx_mock.cpp, -m "not gpu" suite): the host logic under test is the same.
#include <cstdio>
#include <cstdlib>

#include "helib_amd_keys.hpp"

using namespace helib_amd;

typedef std::vector<long> Poly;

// arithmetic modulo (Phi_m(X), p): schoolbook product / substitution, then the remainder by the monic Phi_m
static std::vector<long> g_phi;   // Phi_m, set by main
static Poly reduce_phi(std::vector<long> full, long p)
{
  const size_t n = g_phi.size() - 1;
  for (size_t i = full.size(); i-- > n;) {
    const long c = full[i] % p;
    if (c)
      for (size_t j = 0; j <= n; j++)
        full[i - n + j] = ((full[i - n + j] - c * (g_phi[j] % p)) % p + p) % p;
  }
  full.resize(n);
  for (auto& v : full)
    v = ((v % p) + p) % p;
  return full;
}
static Poly mul(const Poly& a, const Poly& b, long p)
{
  size_t n = a.size();
  std::vector<long> full(2 * n - 1, 0);
  for (size_t i = 0; i < n; i++) {
    if (a[i] == 0)
      continue;
    for (size_t j = 0; j < n; j++)
      full[i + j] = (full[i + j] + (long)((unsigned __int128)a[i] * (unsigned long)b[j] % (unsigned long)p)) % p;
  }
  return reduce_phi(std::move(full), p);
}
static Poly add(const Poly& a, const Poly& b, long p)
{
  Poly out(a.size());
  for (size_t i = 0; i < a.size(); i++)
    out[i] = (a[i] + b[i]) % p;
  return out;
}
static long g_m = 0;
static Poly rot(const Poly& a, long k, long p)   // f(X) -> f(X^k) modulo Phi_m
{
  std::vector<long> full((size_t)g_m, 0);        // first modulo X^m - 1
  for (size_t i = 0; i < a.size(); i++) {
    size_t e = (size_t)((unsigned __int128)i * (unsigned long)k % (unsigned long)g_m);
    full[e] = (full[e] + a[i]) % p;
  }
  return reduce_phi(std::move(full), p);
}
static std::vector<double> mul_d(const std::vector<double>& a, const std::vector<double>& b)
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
static std::vector<double> rot_d(const std::vector<double>& a, long k)
{
  size_t n = a.size();
  std::vector<double> out(n, 0.0);
  for (size_t i = 0; i < n; i++) {
    size_t e = (size_t)((unsigned __int128)i * (unsigned long)k % (2 * n));
    if (e < n)
      out[e] += a[i];
    else
      out[e - n] -= a[i];
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
#define REQUIRE(c)                                                   \
  do {                                                               \
    if (!(c)) {                                                      \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                      \
    }                                                                \
  } while (0)

static int ckks_main(long m, long bits, bool measure);

int main(int argc, char** argv)
{
  if (argc < 5)
    return 2;
  Ctxt::deferNorms() = getenv("HX_TEST_DEFER_NORMS") != nullptr;   // measured norms read back lazily (LazyLn)
  const long m = atol(argv[1]), p = atol(argv[2]), bits = atol(argv[3]);
  const bool measure = atol(argv[4]) != 0;
  if (p == -1)
    return ckks_main(m, bits, measure);
  try {
    ChainContext cc(m, p, 1, bits, 3);
    g_m = m;
    g_phi = cyclotomic(m);
    REQUIRE((long)g_phi.size() == cc.phim + 1 && g_phi.back() == 1);
    long k = 2;                            // the rotation the tests use: the least k > 1 in Zm*
    while (std::gcd(k, m) != 1)
      k++;
    const long k2 = k * k % m, k3 = k2 * k % m;
    auto dev = cc.makeDeviceContext(0);
    SecKey sk(cc, *dev, 4242);
    sk.GenSecKey(3);                       // s^2 -> s and s^3 -> s
    sk.GenKeySWmatrix(1, k);
    sk.setKeySwitchMap();
    REQUIRE(sk.keys.relin && sk.keys.pow.count(3) && sk.keys.automorph.count(k));
    REQUIRE(sk.keys.matrixFor(SKHandle{3, 1}) && !sk.keys.matrixFor(SKHandle{4, 1}) && !sk.keys.matrixFor(SKHandle{2, 3}));
    {
      SKHandle h;
      REQUIRE(h.mul(SKHandle{1, 1}, SKHandle{2, 1}) && h.powerOfS == 3 && h.powerOfX == 1);
      REQUIRE(h.mul(SKHandle{0, 1}, SKHandle{1, 5}) && h.powerOfS == 1 && h.powerOfX == 5);
      REQUIRE(!h.mul(SKHandle{1, 3}, SKHandle{1, 5}));
    }
    const size_t n = (size_t)cc.phim;
    std::mt19937_64 rng(11);
    auto draw = [&]() {
      Poly v(n);
      for (auto& x : v)
        x = (long)(rng() % (uint64_t)p);
      return v;
    };
    auto enc = [&](const Poly& v) {
      Ctxt c = sk.Encrypt(v);
      c.measure = measure;
      return c;
    };
    std::vector<Poly> msg;
    std::vector<Ctxt> ct;
    for (int i = 0; i < 5; i++) {
      msg.push_back(draw());
      ct.push_back(enc(msg.back()));
    }
    const Poly &a = msg[0], &b = msg[1], &c = msg[2];
    const Poly ab = mul(a, b, p), abc = mul(ab, c, p);

    // --- three-way products: parts up to s^3, ONE relinearisation through keySwitchPart
    {
      Ctxt x = ct[0];
      const double cap0 = x.capacity();
      REQUIRE(x.isCorrect() && x.bitCapacity() > 0);
      x.multiplyBy2(ct[1], ct[2]);
      REQUIRE(x.parts.size() == 2 && sk.Decrypt(x) == abc);
      REQUIRE(x.capacity() < cap0 && x.isCorrect());
      Ctxt low = ct[0];                    // the 4-part ciphertext itself decrypts too
      low.multLowLvl(ct[1]);
      low.multLowLvl(ct[2]);
      REQUIRE(low.parts.size() == 4 && low.parts.count(SKHandle{3, 1}) && sk.Decrypt(low) == abc);
      Ctxt y = ct[0];
      y.cube();
      REQUIRE(sk.Decrypt(y) == mul(mul(a, a, p), a, p));
      Ctxt z = ct[0];
      z.square();
      REQUIRE(sk.Decrypt(z) == mul(a, a, p));
      Ctxt w = ct[1];
      w.power(5);
      Poly b2 = mul(b, b, p), b4 = mul(b2, b2, p);
      REQUIRE(sk.Decrypt(w) == mul(b4, b, p));
      Ctxt v = ct[1];
      v.power(4);
      REQUIRE(sk.Decrypt(v) == b4);
    }
    // --- products of many
    {
      Poly all = msg[0];
      for (int i = 1; i < 5; i++)
        all = mul(all, msg[(size_t)i], p);
      Ctxt t = totalProduct(ct);
      REQUIRE(sk.Decrypt(t) == all);
      std::vector<Ctxt> inc(ct.begin(), ct.begin() + 4);
      incrementalProduct(inc);
      Poly run = msg[0];
      REQUIRE(sk.Decrypt(inc[0]) == run);
      for (int i = 1; i < 4; i++) {
        run = mul(run, msg[(size_t)i], p);
        REQUIRE(sk.Decrypt(inc[(size_t)i]) == run);
      }
      std::vector<Ctxt> v1{ct[0], ct[1]}, v2{ct[2], ct[3]};
      Ctxt ip = innerProduct(v1, v2);
      REQUIRE(ip.parts.size() == 2);
      REQUIRE(sk.Decrypt(ip) == add(mul(msg[0], msg[2], p), mul(msg[1], msg[3], p), p));
    }
    // --- hoisted rotations: digits broken once, one key switch per rotation
    {
      Ctxt x = ct[0];
      x.multiplyBy(ct[1]);
      BasicAutomorphPrecon pre(x);
      Ctxt r1 = pre.automorph(1), r3 = pre.automorph(k), r9 = pre.automorph(k2), r27 = pre.automorph(k3);
      REQUIRE(sk.Decrypt(r1) == ab);
      REQUIRE(r3.parts.size() == 2 && sk.Decrypt(r3) == rot(ab, k, p));
      REQUIRE(sk.Decrypt(r9) == rot(ab, k2, p));      // first step hoisted, second by smartAutomorph
      REQUIRE(sk.Decrypt(r27) == rot(ab, k3, p));
      Ctxt s3 = x;
      s3.smartAutomorph(k);
      REQUIRE(sk.Decrypt(s3) == sk.Decrypt(r3));
      REQUIRE(std::fabs(s3.lnNoise - r3.lnNoise) < 2.0);   // same bound up to the cleanUp the precon did first
      bool threw = false;
      try {
        pre.automorph(m - 1);
      } catch (const LogicError&) {
        threw = true;
      }
      REQUIRE(threw);
    }
    // --- Frobenius: X -> X^(p^j), j modulo the order of p
    {
      long ord = 1, x = p % m;
      while (x != 1) {
        x = (long)((unsigned __int128)x * (unsigned long)(p % m) % (unsigned long)m);
        ord++;
      }
      Ctxt f = ct[0];
      f.frobeniusAutomorph(ord);            // the identity
      REQUIRE(sk.Decrypt(f) == a);
      if (ord > 1 && sk.keys.isReachable(p % m)) {
        f.frobeniusAutomorph(1);
        REQUIRE(sk.Decrypt(f) == rot(a, p % m, p));
        f.frobeniusAutomorph(ord + 1);
        REQUIRE(sk.Decrypt(f) == rot(rot(a, p % m, p), p % m, p));
      }
    }
    // --- plaintext constants
    {
      IndexSet allp;
      for (size_t i = 0; i < cc.primes.size(); i++)
        allp.push_back((int)i);
      Poly k1 = draw(), k2 = draw();
      auto balanced = [&](const Poly& v) {
        Poly o(v);
        for (auto& t : o)
          if (t > p / 2)
            t -= p;
        return o;
      };
      DoubleCRT d1 = sk.fromCoeffs(allp, balanced(k1)), d2 = sk.fromCoeffs(allp, balanced(k2));
      Ctxt x = ct[0];
      x.multiplyBy(ct[1]);                  // intFactor != 1 for p > 2: addConstant has to follow it
      x.multByConstant(d1);
      REQUIRE(sk.Decrypt(x) == mul(ab, k1, p));
      x.addConstant(d2);
      REQUIRE(sk.Decrypt(x) == add(mul(ab, k1, p), k2, p));
      Ctxt y = ct[2];
      const double before = y.lnNoise;
      y.multByConstant(5L);
      Poly five(n, 0);
      five[0] = 5 % p;
      REQUIRE(sk.Decrypt(y) == mul(c, five, p));
      if (std::gcd(5L, p) == 1)
        REQUIRE(y.lnNoise == before);       // a unit only moves intFactor
      y.multByConstant(p);                  // 0 modulo the plaintext space: the empty ciphertext
      REQUIRE(y.parts.empty());
      Ctxt s = ct[3], t = ct[4];
      s += t;
      REQUIRE(sk.Decrypt(s) == add(msg[3], msg[4], p));
      s -= t;
      REQUIRE(sk.Decrypt(s) == msg[3]);
      s *= t;
      REQUIRE(sk.Decrypt(s) == mul(msg[3], msg[4], p));
    }
    dev->sync();
  } catch (const std::exception& ex) {
    fprintf(stderr, "exception: %s\n", ex.what());
    return 1;
  }
  printf("ctxt_ops_test OK\n");
  return 0;
}

static int ckks_main(long m, long bits, bool measure)
{
  try {
    const long precision = 20;
    ChainContext cc(m, -1, precision, bits, 3, 3.2, 10.0, 0, 3, 0, true);
    auto dev = cc.makeDeviceContext(0);
    SecKey sk(cc, *dev, 555);
    sk.GenSecKey(3);
    sk.GenKeySWmatrix(1, m - 1);           // complex conjugation
    sk.GenKeySWmatrix(1, 5);
    sk.setKeySwitchMap();
    const size_t n = (size_t)cc.phim;
    const double f = std::ldexp(1.0, (int)precision);
    std::mt19937_64 rng(5);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    auto draw = [&](Poly& scaled, std::vector<double>& enc) {
      scaled.resize(n);
      enc.resize(n);
      for (size_t i = 0; i < n; i++) {
        scaled[i] = std::lround(U(rng) / (double)n * f);
        enc[i] = (double)scaled[i] / f;
      }
    };
    Poly pa, pb, pc;
    std::vector<double> a, b, c;
    draw(pa, a), draw(pb, b), draw(pc, c);
    auto enc = [&](const Poly& v) {
      Ctxt x = sk.CKKSencrypt(v, 1.0, f);
      x.measure = measure;
      return x;
    };
    Ctxt ca = enc(pa), cb = enc(pb), c3 = enc(pc);
    auto within = [&](const Ctxt& ct, const std::vector<double>& want) {
      double err = maxdiff(sk.DecryptCKKS(ct), want), bound = std::exp(ct.lnNoise - ct.lnRatFactor);
      if (!(err <= bound))
        fprintf(stderr, "error %g above the reported bound %g\n", err, bound);
      return err <= bound;
    };
    REQUIRE(within(ca, a) && ca.isCorrect() && ca.capacity() > 0);
    {
      Ctxt x = ca;
      x.multiplyBy2(cb, c3);
      REQUIRE(x.parts.size() == 2 && within(x, mul_d(mul_d(a, b), c)));
      Ctxt y = ca;
      y.cube();
      REQUIRE(within(y, mul_d(mul_d(a, a), a)));
    }
    {  // conjugation and a hoisted rotation
      Ctxt x = ca;
      x.frobeniusAutomorph(1);
      REQUIRE(within(x, rot_d(a, m - 1)));
      x.frobeniusAutomorph(2);              // even: nothing
      REQUIRE(within(x, rot_d(a, m - 1)));
      Ctxt y = ca;
      y.multiplyBy(cb);
      BasicAutomorphPrecon pre(y);
      Ctxt r5 = pre.automorph(5), r25 = pre.automorph(25);
      std::vector<double> ab = mul_d(a, b);
      REQUIRE(within(r5, rot_d(ab, 5)) && within(r25, rot_d(ab, 25)));
    }
    {  // constants: a scalar costs nothing; an encoded constant multiplies the factor
      Ctxt x = ca;
      x.multByConstantCKKS(-2.5);
      std::vector<double> want(n);
      for (size_t i = 0; i < n; i++)
        want[i] = -2.5 * a[i];
      REQUIRE(x.ptxtMag == 2.5 && within(x, want));
      IndexSet allp;
      for (size_t i = 0; i < cc.primes.size(); i++)
        allp.push_back((int)i);
      // the constant polynomial k(X) = 0.75 - 0.5 X scaled by 2^10
      const double kf = 1024.0;
      Poly kc(n, 0);
      kc[0] = (long)(0.75 * kf);
      kc[1] = (long)(-0.5 * kf);
      std::vector<double> kd(n, 0.0);
      kd[0] = 0.75, kd[1] = -0.5;
      DoubleCRT dk = sk.fromCoeffs(allp, kc);
      Ctxt y = ca;
      y.multByConstantCKKS(dk, 1.25, kf, 0.0);
      REQUIRE(within(y, mul_d(a, kd)));
      // adding it: the factor of a fresh ciphertext is an exact multiple of 2^precision only when ef = 1;
      // encode the constant at the ciphertext's own factor instead
      Ctxt z = ca;
      const double zf = std::exp(z.lnRatFactor);
      Poly zc(n, 0);
      zc[0] = std::lround(0.75 * zf);
      zc[1] = std::lround(-0.5 * zf);
      DoubleCRT dz = sk.fromCoeffs(allp, zc);
      z.addConstantCKKS(dz, 1.25, zf);
      for (size_t i = 0; i < n; i++)
        want[i] = a[i] + kd[i];
      REQUIRE(maxdiff(sk.DecryptCKKS(z), want) < 4.0 / f);
    }
    dev->sync();
  } catch (const std::exception& ex) {
    fprintf(stderr, "exception: %s\n", ex.what());
    return 1;
  }
  printf("ctxt_ops_test OK\n");
  return 0;
}

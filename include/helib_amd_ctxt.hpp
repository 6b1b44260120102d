// helib_amd_ctxt.hpp -- C++17 host side of the ciphertext-level path over the C ABI: the
// reference's own control flow for Ctxt::multiplyBy / addCtxt / smartAutomorph with every
// polynomial operation delegated to the engine (helib_amd.h).  Header-only, no NTL.
//
//   PrimeGenerator        src/PrimeGenerator.h:41-126
//   ChainContext          Context::buildModChain (src/Context.cpp:728-1073): small / ctxt / special
//                         primes and digits; noise-bound helpers (include/helib/Context.h)
//   ModuliSizes           src/primeChain.cpp:68-335 (getSet4Size, one- and two-ciphertext forms)
//   Ctxt                  src/Ctxt.cpp: modUpToSet :346-371, modDownToSet :393-562, bringToSet
//                         :373-389, dropSmallAndSpecialPrimes :589-662, addCtxt :1540-1553 (equal
//                         prime sets), tensorProduct :1563-1608, computeIntervalForMul :1610-1656,
//                         multLowLvl / multiplyBy :1681-1774, reLinearize / keySwitchPart :720-842,
//                         automorph / smartAutomorph :2437-2515, cleanUp :788-797
//
// Noise estimates are natural logarithms of the reference's xdouble bounds.  Added noise is the
// reference's high-probability bound (its `#else` branches, src/Ctxt.cpp:546-559 and
// src/DoubleCRT.cpp:520-529) or, with Ctxt::measure = true on power-of-two m, measured on the
// device as in the reference's default build (hx_*_norms, read back synchronously here; the
// python mirror helib_amd/ctxt.py additionally defers the read-back).
// The same sequence of engine calls as helib_amd/ctxt.py: results are bit-identical to it and to
// the CPU oracle (tests/cpp/ctxt_test.cpp in the -m gpu suite, tests/cpp/chain_test.cpp on the CPU).
#pragma once
#include <algorithm>
#include <cmath>
#include <map>
#include <numeric>
#include <set>

#include "helib_amd.hpp"
#include "helib_amd_timing.hpp"

namespace helib_amd {

constexpr long HELIB_SP_NBITS = 60;

namespace detail {
typedef unsigned __int128 u128;
inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)((u128)a * b % q); }
inline uint64_t powmod(uint64_t a, uint64_t e, uint64_t q)
{
  uint64_t r = 1 % q;
  a %= q;
  while (e) {
    if (e & 1)
      r = mulmod(r, a, q);
    a = mulmod(a, a, q);
    e >>= 1;
  }
  return r;
}
// deterministic Miller-Rabin for n < 2^64 (the reference uses NTL::ProbPrime(cand, 60))
inline bool is_prime(uint64_t n)
{
  if (n < 2)
    return false;
  static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (uint64_t p : small) {
    if (n == p)
      return true;
    if (n % p == 0)
      return false;
  }
  uint64_t d = n - 1;
  int s = 0;
  while ((d & 1) == 0) {
    d >>= 1;
    s++;
  }
  for (uint64_t a : small) {
    uint64_t x = powmod(a, d, n);
    if (x == 1 || x == n - 1)
      continue;
    bool comp = true;
    for (int i = 0; i < s - 1 && comp; i++) {
      x = mulmod(x, x, n);
      if (x == n - 1)
        comp = false;
    }
    if (comp)
      return false;
  }
  return true;
}
inline long divc(long a, long b) { return (a + b - 1) / b; }
inline double logaddexp(double a, double b)
{
  if (a == -INFINITY)
    return b;
  if (b == -INFINITY)
    return a;
  double hi = std::max(a, b), lo = std::min(a, b);
  return hi + std::log1p(std::exp(lo - hi));
}
inline double ln(double x) { return x > 0 ? std::log(x) : -INFINITY; }
}  // namespace detail

// primes p = 2^k*t*m + 1 in [(1-1/8)*2^len, 2^len)
class PrimeGenerator {
public:
  static constexpr long B = 3;
  PrimeGenerator(long len, long m) : len_(len), m_(m)
  {
    if (len < B || len > HELIB_SP_NBITS)
      throw InvalidArgument("PrimeGenerator: len is not in [B, HELIB_SP_NBITS]");
    k_ = 0;
    while ((m_ << k_) <= (1L << (len_ - B)))
      k_++;
    t_ = detail::divc((1L << len_) - 1, m_ << k_);
  }
  long next()
  {
    long upper = detail::divc((1L << len_) - 1, m_ << k_);
    for (;;) {
      t_++;
      if (t_ >= upper) {
        k_--;
        if (k_ < ((m_ % 2 == 0) ? 0 : 1))
          throw RuntimeError("Prime generator ran out of primes");
        t_ = detail::divc((1L << len_) - (1L << (len_ - B)) - 1, m_ << k_);
        upper = detail::divc((1L << len_) - 1, m_ << k_);
      }
      if (t_ % 2 == 0)
        continue;
      long cand = ((t_ * m_) << k_) + 1;
      if (detail::is_prime((uint64_t)cand))
        return cand;
    }
  }

private:
  long len_, m_, k_, t_;
};

using PrimeSet = std::set<int>;
inline PrimeSet toSet(const IndexSet& v) { return PrimeSet(v.begin(), v.end()); }
inline IndexSet toVec(const PrimeSet& s) { return IndexSet(s.begin(), s.end()); }
inline PrimeSet operator|(const PrimeSet& a, const PrimeSet& b)
{
  PrimeSet r = a;
  r.insert(b.begin(), b.end());
  return r;
}
inline PrimeSet operator&(const PrimeSet& a, const PrimeSet& b)
{
  PrimeSet r;
  for (int i : a)
    if (b.count(i))
      r.insert(i);
  return r;
}
inline PrimeSet operator-(const PrimeSet& a, const PrimeSet& b)
{
  PrimeSet r;
  for (int i : a)
    if (!b.count(i))
      r.insert(i);
  return r;
}

class ChainContext;

// src/primeChain.cpp:68-335
class ModuliSizes {
public:
  struct Entry {
    double size;
    PrimeSet set;
  };
  void init(const ChainContext& c);
  // getSet4Size(low, high, from[, from2], reverse)
  PrimeSet getSet4Size(double low, double high, const PrimeSet& from1, const PrimeSet* from2, bool reverse) const
  {
    auto cost1 = [&](const PrimeSet& frm, const PrimeSet& to) {
      long c = 100 * (long)(to - frm).size();
      if (iFFT_cost_)
        c += iFFT_cost_ * (long)(frm - to).size();
      return c;
    };
    auto cost = [&](const PrimeSet& s) { return cost1(from1, s) + (from2 ? cost1(*from2, s) : 0); };
    const long n = (long)sizes_.size();
    long idx = (long)(std::lower_bound(sizes_.begin(), sizes_.end(), low,
                                       [](const Entry& e, double v) { return e.size < v; }) -
                      sizes_.begin());
    long best = -1, best_cost = 0, ii = idx;
    while (ii < n && sizes_[ii].size <= high) {
      long c = cost(sizes_[ii].set);
      if (best < 0 || c <= best_cost) {
        best = ii;
        best_cost = c;
      }
      ii++;
    }
    if (from2) {   // src/primeChain.cpp:288-289 / :207-208
      HELIB_AMD_STATS_UPDATE("window2-out", best == -1);
      HELIB_AMD_STATS_UPDATE("window2-nchoices", (double)(ii - idx));
    } else {
      HELIB_AMD_STATS_UPDATE("window1-out", best == -1);
      HELIB_AMD_STATS_UPDATE("window1-nchoices", (double)(ii - idx));
    }
    if (best == -1) {
      const double LN2 = std::log(2.0);
      if (reverse) {
        if (ii < n) {
          double ub = sizes_[ii].size + LN2;
          for (long i = ii; i < n && sizes_[i].size <= ub; i++) {
            long c = cost(sizes_[i].set);
            if (best < 0 || c < best_cost) {
              best = i;
              best_cost = c;
            }
          }
        }
      } else if (idx > 0) {
        double lb = sizes_[idx - 1].size - LN2;
        for (long i = idx - 1; i >= 0 && sizes_[i].size >= lb; i--) {
          long c = cost(sizes_[i].set);
          if (best < 0 || c < best_cost) {
            best = i;
            best_cost = c;
          }
        }
      }
    }
    return best < 0 ? PrimeSet() : sizes_[best].set;
  }
  size_t count() const { return sizes_.size(); }

private:
  std::vector<Entry> sizes_;
  long iFFT_cost_ = 0;
};

// ContextBuilder<BGV>().m(m).p(p).r(r).bits(bits).c(c) -> buildModChain
class ChainContext {
public:
  long m, p, r, ptxtSpace, phim, hwt;
  bool pow2, ckks;   // ckks: ContextBuilder<CKKS> -- p = -1, plaintext space 1, r = precision in bits
  double stdev, scale;
  std::vector<uint64_t> primes;  // Context::moduli order: small, ctxt, special
  IndexSet smallPrimes, ctxtPrimes, specialPrimes;
  std::vector<IndexSet> digits;
  ModuliSizes modSizes;

  ChainContext(long m_, long p_, long r_ = 1, long bits = 300, long c = 3, double stdev_ = 3.2,
               double scale_ = 10.0, long skHwt = 0, long resolution = 3, long bitsInSpecialPrimes = 0,
               bool ckks_ = false)
      : m(m_), p(ckks_ ? -1 : p_), r(r_), hwt(skHwt), ckks(ckks_), stdev(stdev_), scale(scale_)
  {
    ptxtSpace = 1;
    for (long i = 0; i < r && !ckks; i++)
      ptxtSpace *= p;
    phim = eulerPhi(m);
    pow2 = (m & (m - 1)) == 0;
    long pSize = ctxtPrimeSize(bits);
    addSmallPrimes(resolution, pSize);
    addCtxtPrimes(bits, pSize);
    addSpecialPrimes(c, bitsInSpecialPrimes);
    modSizes.init(*this);
  }
  // the device context with Context::moduli registered in order (root 0 = FindPrimRootT; pass the
  // NTL roots for bit-compatibility with an NTL build)
  std::unique_ptr<Context> makeDeviceContext(int device = 0, const std::vector<uint64_t>* roots = nullptr) const
  {
    auto ctx = std::make_unique<Context>((uint64_t)m, device);
    for (size_t i = 0; i < primes.size(); i++)
      ctx->addPrime(primes[i], roots ? (*roots)[i] : 0);
    return ctx;
  }

  double logOfPrime(int i) const { return std::log((double)primes[(size_t)i]); }
  template <class S>
  double logOfProduct(const S& s) const
  {
    double t = 0;
    for (int i : s)
      t += logOfPrime(i);
    return t;
  }
  template <class S>
  uint64_t productOfPrimesMod(const S& s, uint64_t mod) const
  {
    uint64_t t = 1 % mod;
    for (int i : s)
      t = detail::mulmod(t, primes[(size_t)i] % mod, mod);
    return t;
  }
  double noiseBoundForUniform(double magBound, long degBound) const
  {
    return scale * std::sqrt(degBound / 3.0) * magBound;
  }
  double noiseBoundForMod(long modulus, long degBound) const
  {
    double var = (double)modulus * modulus / 12.0 + (modulus % 2 == 0 ? 1.0 / 6.0 : 0.0);
    return scale * std::sqrt(degBound * var);
  }
  double skBound() const
  {
    return hwt > 0 ? std::sqrt(hwt * std::log((double)phim)) : std::sqrt(phim * std::log((double)phim) / 2.0);
  }
  double gaussBound() const
  {
    double eff = pow2 ? std::sqrt(phim * std::log((double)phim)) : std::sqrt(m * std::log((double)phim));
    return (pow2 ? stdev : stdev * std::sqrt((double)m)) * eff;
  }
  // noiseBound of PubKey::Encrypt output (src/keys.cpp:395-475)
  double freshNoiseBound() const
  {
    double e = gaussBound() * ptxtSpace;
    double r_bound = std::sqrt(phim * std::log((double)phim) / 2.0);
    return r_bound * e + e + e * skBound() + noiseBoundForMod(ptxtSpace, phim);
  }
  static long eulerPhi(long n)
  {
    long res = n, x = n;
    for (long q = 2; q * q <= x; q++)
      if (x % q == 0) {
        while (x % q == 0)
          x /= q;
        res -= res / q;
      }
    if (x > 1)
      res -= res / x;
    return res;
  }

private:
  static double bitLoss() { return -std::log1p(-1.0 / (1 << PrimeGenerator::B)) / std::log(2.0); }
  long ctxtPrimeSize(long nBits) const
  {
    double bl = bitLoss();
    long nPrimes = (long)std::ceil(nBits / (HELIB_SP_NBITS - bl));
    long t = HELIB_SP_NBITS;
    while (10 * (t - 1) >= 9 * HELIB_SP_NBITS && (t - 1) >= 30 && ((t - 1) - bl) * nPrimes >= nBits)
      t--;
    return t;
  }
  void add(long q, IndexSet& where)
  {
    if (std::find(primes.begin(), primes.end(), (uint64_t)q) != primes.end())
      throw RuntimeError("Prime q is already in the prime chain");
    primes.push_back((uint64_t)q);
    where.push_back((int)primes.size() - 1);
  }
  void addSmallPrimes(long resolution, long cpSize)
  {
    if (resolution < 1 || resolution > 10)
      resolution = 3;
    std::vector<long> sizes;
    long smallest;
    if (cpSize >= 54)
      smallest = detail::divc(2 * cpSize, 3);
    else if (cpSize >= 45)
      smallest = detail::divc(7 * cpSize, 10);
    else {
      smallest = detail::divc(11 * cpSize, 15);
      sizes.push_back(smallest);
    }
    sizes.push_back(smallest);
    sizes.push_back(smallest);
    for (long delta = resolution; cpSize - delta > smallest; delta *= 2)
      sizes.push_back(cpSize - delta);
    if (cpSize - 3 * resolution > smallest)
      sizes.push_back(cpSize - 3 * resolution);
    if (resolution == 1 && cpSize - 11 > smallest)
      sizes.push_back(cpSize - 11);
    std::sort(sizes.begin(), sizes.end());
    long last = 0;
    std::unique_ptr<PrimeGenerator> gen;
    for (long sz : sizes) {
      if (sz != last)
        gen = std::make_unique<PrimeGenerator>(sz, m);
      add(gen->next(), smallPrimes);
      last = sz;
    }
  }
  void addCtxtPrimes(long nBits, long targetSize)
  {
    PrimeGenerator gen(targetSize, m);
    double bitlen = 0;
    while (bitlen < nBits - 0.5) {
      long q = gen.next();
      add(q, ctxtPrimes);
      bitlen += std::log2((double)q);
    }
  }
  void addSpecialPrimes(long nDgts, long bitsInSpecialPrimes)
  {
    long n = (long)ctxtPrimes.size();
    nDgts = std::max(1L, std::min(nDgts, n));
    digits.clear();
    if (nDgts > 1) {
      IndexSet remaining = ctxtPrimes;
      for (long dgt = 0; dgt < nDgts - 1; dgt++) {
        long card = detail::divc((long)remaining.size(), nDgts - dgt);
        digits.emplace_back(remaining.begin(), remaining.begin() + card);
        remaining.erase(remaining.begin(), remaining.begin() + card);
      }
      if (!remaining.empty())
        digits.push_back(remaining);
    } else {
      digits.push_back(ctxtPrimes);
    }
    double maxDigitLog = 0;
    for (auto& d : digits)
      maxDigitLog = std::max(maxDigitLog, logOfProduct(d));
    nDgts = (long)digits.size();
    const double LN2 = std::log(2.0);
    double nBits;
    if (bitsInSpecialPrimes) {
      nBits = (double)bitsInSpecialPrimes;
    } else {
      double h = hwt == 0 ? phim / 2.0 : (double)hwt;
      double log_phim = std::max(std::log((double)phim), 1.0);
      double p2e = (double)ptxtSpace;
      if (ckks)  // a smaller noise estimate, to protect precision (src/Context.cpp:957-965)
        nBits = (maxDigitLog + std::log(stdev) + std::log((double)nDgts) - 0.5 * std::log(h)) / LN2;
      else if (pow2)
        nBits = (maxDigitLog + std::log(p2e) + std::log(stdev) + 0.5 * std::log(12.0) + std::log((double)nDgts) -
                 0.5 * std::log(log_phim) - 2 * std::log((double)p) - std::log(h)) / LN2;
      else
        nBits = (maxDigitLog + std::log((double)m) + std::log(p2e) + std::log(stdev) + 0.5 * std::log(12.0) +
                 std::log((double)nDgts) - 0.5 * log_phim - 0.5 * std::log(log_phim) - 2 * std::log((double)p) -
                 std::log(h)) / LN2;
    }
    nBits = std::max(nBits, 1.0);
    double bl = bitLoss();
    long nPrimes = (long)std::ceil(nBits / (HELIB_SP_NBITS - bl));
    long t = HELIB_SP_NBITS;
    while ((t - 1) >= 0.55 * HELIB_SP_NBITS && (t - 1) >= 30 && ((t - 1) - bl) * nPrimes >= nBits)
      t--;
    PrimeGenerator gen(t, m);
    while (nPrimes > 0) {
      long q = gen.next();
      if (std::find(primes.begin(), primes.end(), (uint64_t)q) != primes.end())
        continue;
      add(q, specialPrimes);
      nPrimes--;
    }
  }
};

inline void ModuliSizes::init(const ChainContext& c)
{
  iFFT_cost_ = c.pow2 ? 0 : 20;
  std::vector<Entry> sizes{{0.0, {}}};
  for (int i : c.smallPrimes) {
    double sq = c.logOfPrime(i);
    size_t n = sizes.size();
    for (size_t k = 0; k < n; k++) {
      Entry e = sizes[k];
      e.size += sq;
      e.set.insert(i);
      sizes.push_back(e);
    }
  }
  std::vector<Entry> base = sizes;
  PrimeSet interval;
  double isz = 0;
  for (int i : c.ctxtPrimes) {
    interval.insert(i);
    isz += c.logOfPrime(i);
    for (auto& b : base)
      sizes.push_back({b.size + isz, b.set | interval});
  }
  std::sort(sizes.begin(), sizes.end(), [](const Entry& a, const Entry& b) {
    if (a.size != b.size)
      return a.size < b.size;
    return std::lexicographical_compare(a.set.begin(), a.set.end(), b.set.begin(), b.set.end());
  });
  sizes_ = std::move(sizes);
}

// SKHandle (include/helib/Ctxt.h:74-170) for one secret key: (powerOfS, powerOfX)
struct SKHandle {
  long powerOfS = 0, powerOfX = 1;
  bool isOne() const { return powerOfS == 0; }
  bool isBase() const { return powerOfS == 1 && powerOfX == 1; }
  bool operator<(const SKHandle& o) const
  {
    return powerOfS != o.powerOfS ? powerOfS < o.powerOfS : powerOfX < o.powerOfX;
  }
  bool operator==(const SKHandle& o) const
  {
    return (powerOfS == 0 && o.powerOfS == 0) || (powerOfS == o.powerOfS && powerOfX == o.powerOfX);
  }
};

// the key-switching matrices a ciphertext can reach (PubKey::getKeySWmatrix for this path)
struct KeySet {
  const KeySwitch* relin = nullptr;             // s^2 -> s
  std::map<long, const KeySwitch*> automorph;   // k -> s(X^k) -> s
  long ptxtSpace = 0;                           // KeySwitch::ptxtSpace
  double lnNoise = 0;                           // ln KeySwitch::noiseBound
  std::vector<long> keySwitchMap;               // k -> first step on the way to X -> X^k (0: none)

  // PubKey::setKeySwitchMap (src/keys.cpp:122-172): BFS over Zm* from 1 along the available matrices
  void setKeySwitchMap(long m)
  {
    keySwitchMap.assign((size_t)m, 0);
    std::vector<long> queue{1};
    for (size_t head = 0; head < queue.size(); head++) {
      long cur = queue[head];
      for (auto& kv : automorph) {
        long nxt = (long)((unsigned __int128)cur * (unsigned long)kv.first % (unsigned long)m);
        if (nxt != 1 && keySwitchMap[(size_t)nxt] == 0) {
          keySwitchMap[(size_t)nxt] = kv.first;
          queue.push_back(nxt);
        }
      }
    }
  }
  bool isReachable(long k) const { return k == 1 || firstStep(k) != 0; }
  // PubKey::getNextKSWmatrix(k).fromKey.getPowerOfX()
  long firstStep(long k) const
  {
    if (!keySwitchMap.empty())
      return keySwitchMap[(size_t)k];
    return automorph.count(k) ? k : 0;
  }
};

class Ctxt {
public:
  static constexpr double safety = 0.6931471805599453;  // ln 2, src/Ctxt.cpp:39
  bool measure = false;  // measured added noise (hx_*_norms) instead of the bounds; power-of-two m

  const ChainContext* context;
  const Context* dev;
  const KeySet* keys;
  std::map<SKHandle, DoubleCRT> parts;
  PrimeSet primeSet;
  long ptxtSpace, intFactor = 1;
  double lnNoise = -INFINITY;
  double ptxtMag = 1.0, lnRatFactor = 0.0;  // CKKS: |plaintext| bound, ln of the scaling factor

  Ctxt(const ChainContext& c, const Context& d, const KeySet& k) : context(&c), dev(&d), keys(&k), ptxtSpace(c.ptxtSpace) {}
  // a fresh 2-part ciphertext over the ctxt primes (noise bound of PubKey::Encrypt)
  static Ctxt fresh(const ChainContext& c, const Context& d, const KeySet& k, DoubleCRT c0, DoubleCRT c1)
  {
    Ctxt ct(c, d, k);
    ct.parts.emplace(SKHandle{0, 1}, std::move(c0));
    ct.parts.emplace(SKHandle{1, 1}, std::move(c1));
    ct.primeSet = toSet(c.ctxtPrimes);
    ct.lnNoise = std::log(c.freshNoiseBound());
    return ct;
  }

  double logOfPrimeSet() const { return context->logOfProduct(primeSet); }
  double modSwitchAddedNoiseBound() const
  {
    double h = context->skBound(), added = 0;
    for (auto& kv : parts)
      added += std::pow(h, (double)kv.first.powerOfS);
    return added * context->noiseBoundForUniform(ptxtSpace / 2.0, context->phim);
  }

  // ---- prime-set maintenance ----
  void modUpToSet(const PrimeSet& s)
  {
    HELIB_AMD_TIMER_START;
    PrimeSet diff = s - primeSet;
    if (diff.empty())
      return;
    IndexSet d = toVec(diff);
    for (auto& kv : parts)
      kv.second.addPrimesAndScale(d);
    lnNoise += context->logOfProduct(diff);
    lnRatFactor += context->logOfProduct(diff);  // "If CKKS, the rational factor grows" (:366)
    primeSet = primeSet | diff;
  }
  void modDownToSet(const PrimeSet& s)
  {
    HELIB_AMD_TIMER_START;
    PrimeSet inter = primeSet & s;
    if (inter.empty())
      throw RuntimeError("modDownToSet called with disjoint sets");
    PrimeSet diff = primeSet - inter;
    if (diff.empty())
      return;
    std::vector<Ctxt*> one{this};
    const double addedBound = modSwitchAddedNoiseBound();
    std::vector<double> added = modDownParts(one, inter, PrimeSet());
    HELIB_AMD_STATS_UPDATE("mod-switch-added-noise", added[0] / addedBound);   // src/Ctxt.cpp:535-537
    lnNoise = detail::logaddexp(lnNoise - context->logOfProduct(diff), detail::ln(added[0]));
    lnRatFactor -= context->logOfProduct(diff);  // ratFactor /= f (:533, :553)
    primeSet = inter;
  }
  void bringToSet(const PrimeSet& s0)
  {
    HELIB_AMD_TIMER_START;
    PrimeSet s = s0.empty() ? PrimeSet{context->ctxtPrimes[0]} : s0;
    std::vector<Ctxt*> one{this};
    bringManyToSet(one, s);
  }
  void dropSmallAndSpecialPrimes()
  {
    HELIB_AMD_TIMER_START;
    PrimeSet small = toSet(context->smallPrimes), ctp = toSet(context->ctxtPrimes);
    if ((primeSet & small).empty()) {
      modDownToSet(ctp);
      return;
    }
    PrimeSet target = primeSet & ctp;
    PrimeSet dropping = primeSet - target;
    double log_dropping = context->logOfProduct(dropping);
    double log_msn = std::log(modSwitchAddedNoiseBound()) + 3 * std::log(2.0);
    double comp = 0;
    if (lnNoise - log_dropping + comp < log_msn) {
      for (int i : ctp - target) {
        target.insert(i);
        comp += context->logOfPrime(i);
        if (lnNoise - log_dropping + comp >= log_msn)
          break;
      }
    }
    bringToSet(target);
  }

  // ---- arithmetic ----
  // Ctxt::relin_CKKS_adjust (src/Ctxt.cpp:664-717): if the noise is below what the special primes
  // were sized for, scale the ciphertext (and its factor) up by an integer
  void relin_CKKS_adjust()
  {
    if (!context->ckks)
      return;
    double h = context->hwt == 0 ? context->phim / 2.0 : (double)context->hwt;
    double log_phim = std::max(std::log((double)context->phim), 1.0);
    double lnGamma = std::log(8.0 * (double)(long)context->scale * std::sqrt(context->phim * log_phim * h / 12.0));
    if (lnGamma > lnNoise) {
      long xf = (long)std::ceil(std::exp(lnGamma - lnNoise));
      for (auto& kv : parts)
        kv.second *= xf;
      lnNoise += std::log((double)xf);
      lnRatFactor += std::log((double)xf);
    }
  }
  // Ctxt::mulIntFactor (src/Ctxt.cpp:331-340)
  void mulIntFactor(long e)
  {
    if (e == 1)
      return;
    intFactor = (long)detail::mulmod((uint64_t)intFactor, (uint64_t)e, (uint64_t)ptxtSpace);
    long bal = e > ptxtSpace / 2 ? e - ptxtSpace : e;
    for (auto& kv : parts)
      kv.second *= bal;
    lnNoise += std::log((double)std::labs(bal));
  }
  void negate()
  {
    for (auto& kv : parts)
      kv.second.Negate();
  }
  // Ctxt::equalizeRationalFactors (src/Ctxt.cpp:1212-1356): bring two CKKS ciphertexts to one factor
  // by small integer multipliers (continued-fraction convergents of the ratio), stopping once the
  // discretisation error is within sqrt(2) of the error the sum has anyway.  Relative to the smaller
  // factor, so doubles do (the reference's xdouble carries the same 53 bits).
  static void equalizeRationalFactors(Ctxt& c1, Ctxt& c2)
  {
    Ctxt& big = c1.lnRatFactor > c2.lnRatFactor ? c1 : c2;
    Ctxt& small = c1.lnRatFactor > c2.lnRatFactor ? c2 : c1;
    const double base = small.lnRatFactor, x = std::exp(big.lnRatFactor - base);
    const double denomBound = std::ldexp(1.0, (int)c1.context->r + 1), epsilon = 0.125 / denomBound;
    double a = std::floor(x + epsilon), xi = x - a;
    double prevDenom = 0, denom = 1, numer = std::floor(denom * x + 0.5);
    const double m1 = big.ptxtMag, of1 = x, oe1 = std::exp(big.lnNoise - base);
    const double m2 = small.ptxtMag, of2 = 1.0, oe2 = std::exp(small.lnNoise - base);
    const double target = oe1 / of1 + oe2 / of2;
    double f, fe1, fe2;
    for (;;) {
      double f1 = of1 * denom, e1 = oe1 * denom, f2 = of2 * numer, e2 = oe2 * numer;
      auto calc = [&](double ff) { return m1 * std::fabs(f1 / ff - 1.0) + m2 * std::fabs(f2 / ff - 1.0) + (e1 + e2) / ff; };
      double err1 = calc(f1), err2 = calc(f2), err;
      if (err1 < err2) {
        f = f1, fe1 = e1, fe2 = e2 + m2 * std::fabs(f2 - f1), err = err1;
      } else {
        f = f2, fe1 = e1 + m1 * std::fabs(f2 - f1), fe2 = e2, err = err2;
      }
      if (err < std::sqrt(2.0) * target || xi <= 0)
        break;
      xi = 1.0 / xi;
      double ai = std::floor(xi + epsilon);
      xi -= ai;
      double tmpDenom = denom * ai + prevDenom;
      if (tmpDenom > denomBound)
        break;
      prevDenom = denom;
      denom = tmpDenom;
      numer = std::floor(denom * x + 0.5);
    }
    if (denom != 1)
      for (auto& kv : big.parts)
        kv.second *= (long)denom;
    if (numer != 1)
      for (auto& kv : small.parts)
        kv.second *= (long)numer;
    big.lnRatFactor = small.lnRatFactor = std::log(f) + base;
    big.lnNoise = detail::ln(fe1) + base;
    small.lnNoise = detail::ln(fe2) + base;
  }
  // Ctxt::addCtxt (src/Ctxt.cpp:1405-1556): plaintext spaces reduced to their gcd (BGV), both
  // operands mod-switched UP to the union of their prime sets, CKKS factors equalised, BGV
  // intFactors harmonised by the (e1, e2) of least noise along the extended Euclidean sequence,
  // then the parts added handle by handle.
  void addCtxt(const Ctxt& other, bool negative = false)
  {
    HELIB_AMD_TIMER_START;
    if (other.parts.empty())
      return;
    if (parts.empty()) {
      *this = other;
      if (negative)
        negate();
      return;
    }
    const Ctxt* o = &other;
    std::unique_ptr<Ctxt> tmp;
    auto own = [&]() -> Ctxt& {
      if (!tmp) {
        tmp = std::make_unique<Ctxt>(other);
        o = tmp.get();
      }
      return *tmp;
    };
    if (context->ckks) {
      if (ptxtSpace != 1 || other.ptxtSpace != 1)
        throw RuntimeError("Plaintext spaces incompatible");
    } else {
      long g = std::gcd(ptxtSpace, other.ptxtSpace);
      if (g <= 1)
        throw RuntimeError("New and old plaintext spaces are coprime");
      ptxtSpace = g;
      intFactor %= g;
      if (other.ptxtSpace != g) {
        own().ptxtSpace = g;
        tmp->intFactor %= g;
      }
    }
    if (!(o->primeSet - primeSet).empty())
      modUpToSet(primeSet | o->primeSet);
    if (!(primeSet - o->primeSet).empty())
      own().modUpToSet(primeSet);
    if (context->ckks)
      equalizeRationalFactors(*this, own());
    long e1 = 1, e2 = 1;
    if (!context->ckks && intFactor != o->intFactor) {
      const long P = ptxtSpace;
      auto inv = [&](long v) {  // v^-1 mod P
        long aa = v, bb = P, x0 = 1, x1 = 0;
        while (bb) {
          long q = aa / bb, t = aa % bb;
          aa = bb, bb = t;
          t = x0 - q * x1, x0 = x1, x1 = t;
        }
        return ((x0 % P) + P) % P;
      };
      long ratio = (long)detail::mulmod((uint64_t)o->intFactor, (uint64_t)inv(intFactor), (uint64_t)P);
      auto bal = [&](long e) { return (double)std::labs(e > P / 2 ? e - P : e); };
      auto norm = [&](long ea, long eb) {
        return detail::logaddexp(lnNoise + detail::ln(bal(ea)), o->lnNoise + detail::ln(bal(eb)));
      };
      long r0 = P, t0 = 0, r1 = ratio, t1 = 1;
      e1 = r1, e2 = t1;
      double best = norm(e1, e2);
      while (r1 != 0) {
        long q = r0 / r1, r2 = r0 % r1, t2 = t0 - t1 * q;
        r0 = r1, r1 = r2, t0 = t1, t1 = t2;
        long ea = ((r1 % P) + P) % P, eb = ((t1 % P) + P) % P;
        if (ea % context->p != 0) {
          double cand = norm(ea, eb);
          if (cand < best)
            e1 = ea, e2 = eb, best = cand;
        }
      }
    }
    if (e2 != 1)
      own().mulIntFactor(e2);
    if (e1 != 1)
      mulIntFactor(e1);
    for (auto& kv : o->parts) {
      auto it = parts.find(kv.first);
      if (it == parts.end()) {
        auto ins = parts.emplace(kv.first, kv.second).first;
        if (negative)
          ins->second.Negate();
      } else if (negative) {
        it->second -= kv.second;
      } else {
        it->second += kv.second;
      }
    }
    ptxtMag += o->ptxtMag;
    lnNoise = detail::logaddexp(lnNoise, o->lnNoise);
  }
  // Ctxt::computeIntervalForMul (src/Ctxt.cpp:1610-1656): [lo, hi] = ln of the target modulus size
  static std::pair<double, double> computeIntervalForMul(const Ctxt& c1, const Ctxt& c2);
  // multLowLvl: bring both to a common set, tensor
  void multLowLvl(Ctxt other)
  {
    HELIB_AMD_TIMER_START;
    if (parts.empty() || other.parts.empty()) {
      parts.clear();
      return;
    }
    if (context->ckks) {
      if (ptxtSpace != 1 || other.ptxtSpace != 1)
        throw RuntimeError("Plaintext spaces incompatible");
    } else {
      long g = std::gcd(ptxtSpace, other.ptxtSpace);
      if (g <= 1)
        throw RuntimeError("Plaintext spaces are co-prime");
      ptxtSpace = other.ptxtSpace = g;
      intFactor %= g;
      other.intFactor %= g;
    }
    auto iv = computeIntervalForMul(*this, other);
    PrimeSet s = context->modSizes.getSet4Size(iv.first, iv.second, primeSet, &other.primeSet, context->ckks);
    if (primeSet == other.primeSet) {
      std::vector<Ctxt*> both{this, &other};
      bringManyToSet(both, s.empty() ? PrimeSet{context->ctxtPrimes[0]} : s);
    } else {
      bringToSet(s);
      other.bringToSet(s);
    }
    tensorProduct(other);
  }
  void multiplyBy(const Ctxt& other)
  {
    HELIB_AMD_TIMER_START;
    multLowLvl(other);  // works on a copy of `other`, as the reference does (src/Ctxt.cpp:1716-1745)
    reLinearize();
  }
  void multiplyBy(Ctxt&& other)  // the operand may be consumed: no copy
  {
    HELIB_AMD_TIMER_START;
    multLowLvl(std::move(other));
    reLinearize();
  }
  void reLinearize()
  {
    HELIB_AMD_TIMER_START;
    SKHandle hnd;
    int n_other = 0;
    for (auto& kv : parts)
      if (!kv.first.isOne() && !kv.first.isBase()) {
        hnd = kv.first;
        n_other++;
      }
    if (n_other == 0)
      return;
    if (n_other > 1)
      throw LogicError("one non-canonical part at a time");
    const KeySwitch* W = nullptr;
    if (hnd.powerOfS == 2 && hnd.powerOfX == 1)
      W = keys->relin;
    else if (hnd.powerOfS == 1) {
      auto it = keys->automorph.find(hnd.powerOfX);
      W = it == keys->automorph.end() ? nullptr : it->second;
    }
    if (!W)
      throw LogicError("no key-switching matrices for this part");
    dropSmallAndSpecialPrimes();
    relin_CKKS_adjust();
    const IndexSet& sp = context->specialPrimes;
    double logProd = context->logOfProduct(sp);
    lnRatFactor += logProd;  // the CKKS factor after the mod-up by the special primes (:757)
    std::vector<IndexSet> digits;
    for (auto& d : context->digits) {
      IndexSet r;
      for (int i : d)
        if (primeSet.count(i))
          r.push_back(i);
      if (!r.empty())
        digits.push_back(r);
    }
    if (ptxtSpace > 1) {  // g == 1 for CKKS
      ptxtSpace = std::gcd(ptxtSpace, keys->ptxtSpace ? keys->ptxtSpace : context->ptxtSpace);
      intFactor %= ptxtSpace;
    }
    DoubleCRT& t0 = parts.at(SKHandle{0, 1});
    auto its = parts.find(SKHandle{1, 1});
    DoubleCRT& t2 = parts.at(hnd);
    IndexSet own = t0.getIndexSet();
    DoubleCRT o0(*dev, own, t0.batch(), DoubleCRT::Uninitialized{}), o1(*dev, own, t0.batch(), DoubleCRT::Uninitialized{});
    std::vector<int> idx, off;
    flatten(digits, idx, off);
    std::vector<double> nrm((size_t)digits.size() * (size_t)t0.batch(), 0.0);
    if (measure) {
      check(hx_ctx_defer_norms(dev->handle(), 0));
      check(hx_relinearize_norms(t0.handle(), its == parts.end() ? nullptr : its->second.handle(), t2.handle(),
                                 W->handle(), idx.data(), off.data(), (int)digits.size(), sp.data(), (int)sp.size(),
                                 o0.handle(), o1.handle(), nrm.data()));
    } else {
      check(hx_relinearize(t0.handle(), its == parts.end() ? nullptr : its->second.handle(), t2.handle(),
                           W->handle(), idx.data(), off.data(), (int)digits.size(), sp.data(), (int)sp.size(),
                           o0.handle(), o1.handle()));
    }
    double added = -INFINITY;
    for (size_t k = 0; k < digits.size(); k++) {
      double nb;
      if (measure) {
        double mx = 0;
        for (int b = 0; b < t0.batch(); b++)
          mx = std::max(mx, nrm[k * (size_t)t0.batch() + (size_t)b]);
        nb = detail::ln(mx) + context->logOfProduct(digits[k]);
      } else {
        nb = std::log(context->noiseBoundForUniform(0.5, context->phim)) + context->logOfProduct(digits[k]);
      }
      added = detail::logaddexp(added, nb + keys->lnNoise);
    }
    lnNoise = detail::logaddexp(lnNoise + logProd, added);
    parts.clear();
    parts.emplace(SKHandle{0, 1}, std::move(o0));
    parts.emplace(SKHandle{1, 1}, std::move(o1));
    primeSet = primeSet | toSet(sp);
  }
  // Ctxt::automorph: F(X) -> F(X^k) on every part; handles follow
  void automorph(long k)
  {
    HELIB_AMD_TIMER_START;
    long m = context->m;
    k = ((k % m) + m) % m;
    if (std::gcd(k, m) != 1)
      throw InvalidArgument("k must be in Zm*");
    if (k == 1)
      return;
    std::map<SKHandle, DoubleCRT> np;
    for (auto& kv : parts) {
      kv.second.automorph(k);
      SKHandle h = kv.first;
      if (!h.isOne())
        h.powerOfX = (long)((unsigned __int128)h.powerOfX * (unsigned long)k % (unsigned long)m);
      np.emplace(h, std::move(kv.second));
    }
    parts = std::move(np);
  }
  // Ctxt::smartAutomorph (src/Ctxt.cpp:2462-2515): walk the path of available matrices
  void smartAutomorph(long k)
  {
    HELIB_AMD_TIMER_START;
    long m = context->m;
    k = ((k % m) + m) % m;
    if (k == 1 || parts.empty())
      return;
    if (std::gcd(k, m) != 1)
      throw InvalidArgument("k must be in Zm*");
    if (!keys->isReachable(k))
      throw LogicError("no key-switching matrices for k=" + std::to_string(k));
    reLinearize();
    while (k != 1) {
      long amt = keys->firstStep(k);
      if (amt == 0)
        throw LogicError("no key-switching matrices for k=" + std::to_string(k));
      automorph(amt);
      reLinearize();
      // k *= amt^-1 mod m
      long inv = 1, a = amt % m, e = ChainContext::eulerPhi(m) - 1;
      while (e) {
        if (e & 1)
          inv = (long)((unsigned __int128)inv * (unsigned long)a % (unsigned long)m);
        a = (long)((unsigned __int128)a * (unsigned long)a % (unsigned long)m);
        e >>= 1;
      }
      k = (long)((unsigned __int128)k * (unsigned long)inv % (unsigned long)m);
    }
  }
  void cleanUp()
  {
    reLinearize();
    if (!(primeSet & (toSet(context->specialPrimes) | toSet(context->smallPrimes))).empty())
      dropSmallAndSpecialPrimes();
  }

private:
  void tensorProduct(const Ctxt& o)
  {
    if (parts.size() != 2 || o.parts.size() != 2)
      throw LogicError("tensorProduct: two-part operands expected");
    if (ptxtSpace > 2) {
      uint64_t q = context->productOfPrimesMod(primeSet, (uint64_t)ptxtSpace);
      intFactor = (long)detail::mulmod(detail::mulmod((uint64_t)intFactor, (uint64_t)o.intFactor, (uint64_t)ptxtSpace), q,
                                       (uint64_t)ptxtSpace);
    }
    const DoubleCRT &c0 = parts.at(SKHandle{0, 1}), &c1 = parts.at(SKHandle{1, 1});
    const DoubleCRT &d0 = o.parts.at(SKHandle{0, 1}), &d1 = o.parts.at(SKHandle{1, 1});
    IndexSet idx = c0.getIndexSet();
    DoubleCRT::Uninitialized u;
    DoubleCRT t0(*dev, idx, c0.batch(), u), t1(*dev, idx, c0.batch(), u), t2(*dev, idx, c0.batch(), u);
    helib_amd::tensorProduct(c0, c1, d0, d1, t0, t1, t2);
    parts.clear();
    parts.emplace(SKHandle{0, 1}, std::move(t0));
    parts.emplace(SKHandle{1, 1}, std::move(t1));
    parts.emplace(SKHandle{2, 1}, std::move(t2));
    if (context->ckks) {  // totalNoiseBound = factor*ptxt + noiseBound on both sides (:1600-1606)
      double n1 = lnNoise, n2 = o.lnNoise;
      lnNoise = detail::logaddexp(detail::logaddexp(n1 + detail::ln(o.ptxtMag) + o.lnRatFactor,
                                                    n2 + detail::ln(ptxtMag) + lnRatFactor), n1 + n2);
      lnRatFactor += o.lnRatFactor;
      ptxtMag *= o.ptxtMag;
    } else {
      lnNoise += o.lnNoise;
    }
  }
  // polynomial work of modDownToSet (after a mod-up by `add`) on all parts of ciphertexts that share
  // one prime set: one fused call; returns the added noise per ciphertext
  static std::vector<double> modDownParts(std::vector<Ctxt*>& cts, const PrimeSet& keep, const PrimeSet& add)
  {
    Ctxt& a = *cts[0];
    std::vector<hx_poly*> polys;
    for (Ctxt* c : cts)
      for (auto& kv : c->parts)
        polys.push_back(kv.second.handle());
    PrimeSet cur = a.primeSet | add;
    IndexSet drop = toVec(cur - keep), addv = toVec(add);
    int batch = cts[0]->parts.begin()->second.batch();
    std::vector<double> norms(polys.size() * (size_t)batch, 0.0);
    uint64_t pt = (uint64_t)a.ptxtSpace;
    int rc;
    if (a.measure) {
      check(hx_ctx_defer_norms(a.dev->handle(), 0));
      rc = addv.empty() ? hx_scale_down_multi_norms(polys.data(), (int)polys.size(), drop.data(), (int)drop.size(), pt,
                                                    norms.data(), nullptr)
                        : hx_bring_to_set_multi_norms(polys.data(), (int)polys.size(), addv.data(), (int)addv.size(),
                                                      drop.data(), (int)drop.size(), pt, norms.data());
    } else {
      rc = addv.empty() ? hx_scale_down_multi(polys.data(), (int)polys.size(), drop.data(), (int)drop.size(), pt)
                        : hx_bring_to_set_multi(polys.data(), (int)polys.size(), addv.data(), (int)addv.size(),
                                                drop.data(), (int)drop.size(), pt);
    }
    check(rc);
    std::vector<double> out;
    size_t k = 0;
    double h = a.context->skBound();
    for (Ctxt* c : cts) {
      if (!a.measure) {
        out.push_back(c->modSwitchAddedNoiseBound());
        continue;
      }
      double sum = 0;
      for (auto& kv : c->parts) {
        double mx = 0;
        for (int b = 0; b < batch; b++)
          mx = std::max(mx, norms[k * (size_t)batch + (size_t)b]);
        sum += mx * std::pow(h, (double)kv.first.powerOfS);
        k++;
      }
      out.push_back(sum);
    }
    return out;
  }
  // bringToSet(s) = modUpToSet(s); modDownToSet(s) for ciphertexts on one prime set, all parts in one call
  static void bringManyToSet(std::vector<Ctxt*>& cts, const PrimeSet& s)
  {
    Ctxt& a = *cts[0];
    PrimeSet add = s - a.primeSet, up = a.primeSet | add, inter = up & s;
    if (inter.empty())
      throw RuntimeError("modDownToSet called with disjoint sets");
    PrimeSet diff = up - inter;
    if (add.empty() && diff.empty())
      return;
    std::vector<double> added;
    if (diff.empty()) {  // pure mod-up
      IndexSet d = toVec(add);
      for (Ctxt* c : cts)
        for (auto& kv : c->parts)
          kv.second.addPrimesAndScale(d);
    } else {
      added = modDownParts(cts, inter, add);
    }
    for (size_t i = 0; i < cts.size(); i++) {
      Ctxt* c = cts[i];
      c->lnNoise += c->context->logOfProduct(add);
      c->lnRatFactor += c->context->logOfProduct(add) - c->context->logOfProduct(diff);
      c->primeSet = up;
      if (!diff.empty()) {
        c->lnNoise = detail::logaddexp(c->lnNoise - c->context->logOfProduct(diff), detail::ln(added[i]));
        c->primeSet = inter;
      }
    }
  }
};

inline std::pair<double, double> Ctxt::computeIntervalForMul(const Ctxt& c1, const Ctxt& c2)
{
  const double LN2 = std::log(2.0);
  double cap1 = c1.logOfPrimeSet() - std::max(c1.lnNoise, 0.0);
  double cap2 = c2.logOfPrimeSet() - std::max(c2.lnNoise, 0.0);
  double adn1 = std::log(c1.modSwitchAddedNoiseBound()), adn2 = std::log(c2.modSwitchAddedNoiseBound());
  if (c1.context->ckks) {  // the opposite end: keep n*q'/q above the added noise (:1637-1651)
    double lo = std::max(cap1 + adn1, cap2 + adn2) + safety;
    return {lo, lo + 4 * LN2};
  }
  double hi = std::min(cap1 + adn1, cap2 + adn2) - safety;
  return {hi - 4 * LN2, hi};
}

}  // namespace helib_amd

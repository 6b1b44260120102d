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
#include <cstdlib>
#include <functional>
#include <map>
#include <mutex>
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
    if (bits <= 0)   // Context::buildModChain (src/Context.cpp:1044-1046)
      throw InvalidArgument("Cannot initialise modulus chain with nBits < 1");
    if (skHwt < 0)
      throw InvalidArgument("invalid skHwt parameter");
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

  // ---- size and security of the chain (include/helib/Context.h:857-889, src/Context.cpp:34-72) ----
  long bitSizeOfQ() const
  {
    return (long)std::ceil((logOfProduct(ctxtPrimes) + logOfProduct(specialPrimes)) / std::log(2.0));
  }
  // lweEstimateSecurity: the reference's affine fits to the LWE estimator, slope and constant interpolated
  // between the fitted Hamming weights 120, 150, ..., 450 (dense keys: 3.8, -20); never negative
  static double lweEstimateSecurity(long n, double log2AlphaInv, long hwt)
  {
    constexpr long MIN_SK_HWT = 120;
    if (hwt < 0 || (hwt > 0 && hwt < MIN_SK_HWT))
      return 0.0;
    static const double hw[] = {120, 150, 180, 210, 240, 270, 300, 330, 360, 390, 420, 450};
    static const double sl[] = {2.4, 2.67, 2.83, 3.0, 3.1, 3.3, 3.3, 3.35, 3.4, 3.45, 3.5, 3.55};
    static const double cn[] = {19, 13, 10, 6, 3, 1, -3, -4, -5, -7, -10, -12};
    const size_t nw = sizeof(hw) / sizeof(hw[0]);
    double slope, cst;
    if (hwt == 0) {
      slope = 3.8, cst = -20;
    } else {
      const size_t i = (size_t)((hwt - 120) / 30);
      if (i < nw - 1) {
        const double a = ((double)hwt - hw[i]) / (hw[i + 1] - hw[i]);
        slope = sl[i] + a * (sl[i + 1] - sl[i]);
        cst = cn[i] + a * (cn[i + 1] - cn[i]);
      } else {
        slope = sl[nw - 1], cst = cn[nw - 1];
      }
    }
    const double ret = slope * (double)n / log2AlphaInv + cst;
    return ret < 0.0 ? 0.0 : ret;
  }
  double securityLevel() const
  {
    const double s = pow2 ? stdev : stdev * std::sqrt((double)m);
    const double log2AlphaInv = (logOfProduct(ctxtPrimes) + logOfProduct(specialPrimes) - std::log(s)) / std::log(2.0);
    return lweEstimateSecurity(phim, log2AlphaInv, hwt);
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
  // EncryptedArrayCx::encodeRoundingError / encodeScalingFactor (include/helib/EncryptedArray.h:1287-1312): the
  // factor CKKS plaintexts are scaled by before rounding -- ceil(precision * roundErr) rounded up to a power of
  // two, precision defaulting to 2^r of ContextBuilder<CKKS>::precision(r)
  double encodeRoundingError() const { return noiseBoundForUniform(0.5, phim); }
  long encodeScalingFactor(long precision = -1, double roundErr = -1.0) const
  {
    if (precision <= 0)
      precision = 1L << r;
    if (roundErr < 0)
      roundErr = encodeRoundingError();
    long f = (long)std::ceil((double)precision * roundErr), k = 0;
    while ((1L << k) < f)   // NTL::NextPowerOfTwo
      k++;
    return 1L << k;
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
  // SKHandle::mul (include/helib/Ctxt.h:140-165): powers of s add; the automorphism amounts must agree
  // unless one side is the constant handle
  bool mul(const SKHandle& a, const SKHandle& b)
  {
    if (a.isOne())
      *this = b;
    else if (b.isOne())
      *this = a;
    else if (a.powerOfX != b.powerOfX)
      return false;
    else
      *this = SKHandle{a.powerOfS + b.powerOfS, a.powerOfX};
    return true;
  }
};

// the key-switching matrices a ciphertext can reach (PubKey::getKeySWmatrix for this path)
struct KeySet {
  const KeySwitch* relin = nullptr;             // s^2 -> s
  std::map<long, const KeySwitch*> automorph;   // k -> s(X^k) -> s
  std::map<long, const KeySwitch*> pow;         // e >= 3 -> s^e -> s (GenSecKey(maxDegKswitch >= 3))
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
  // PubKey::getKeySWmatrix(handle) (src/keys.cpp:218-239); nullptr when there is none
  const KeySwitch* matrixFor(const SKHandle& h) const
  {
    if (h.powerOfS == 2 && h.powerOfX == 1)
      return relin;
    if (h.powerOfS == 1) {
      auto it = automorph.find(h.powerOfX);
      return it == automorph.end() ? nullptr : it->second;
    }
    if (h.powerOfX == 1) {
      auto it = pow.find(h.powerOfS);
      return it == pow.end() ? nullptr : it->second;
    }
    return nullptr;
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

// ln(noiseBound) whose measured contributions may still be on the device.  With Ctxt::deferNorms an operation
// enqueues its kernels, registers how its norms will enter the estimate and returns; the value is completed
// (Context::flushNorms waits only for the norm kernels already enqueued) when somebody reads it -- normally the
// next prime-set decision, by which time more work is queued behind on the GPU.  Reads, assignments and += look
// like a double's.  The registered updates see only the value and what they captured (shared norm arrays,
// constants), never the ciphertext, so copies and moves of a Ctxt carry them along.
class LazyLn {
public:
  LazyLn(double v = -INFINITY) : v_(v) {}
  operator double() const
  {
    settle();
    return v_;
  }
  LazyLn& operator=(double x)   // (a value computed from this one has settled it already; an unrelated one replaces it)
  {
    pending_.clear();
    v_ = x;
    return *this;
  }
  LazyLn& operator+=(double x)
  {
    if (pending_.empty())
      v_ += x;
    else
      pending_.push_back([x](double& v) { v += x; });
    return *this;
  }
  // f(v) now, or -- lazy -- when the value is next read
  void apply(const Context* dev, bool lazy, std::function<void(double&)> f)
  {
    if (lazy) {
      dev_ = dev;
      pending_.push_back(std::move(f));
    } else {
      settle();
      f(v_);
    }
  }
  bool pending() const { return !pending_.empty(); }

private:
  void settle() const
  {
    if (pending_.empty())
      return;
    dev_->flushNorms();
    for (auto& f : pending_)
      f(v_);
    pending_.clear();
  }
  mutable double v_;
  mutable std::vector<std::function<void(double&)>> pending_;
  const Context* dev_ = nullptr;
};

class Ctxt {
public:
  static constexpr double safety = 0.6931471805599453;  // ln 2, src/Ctxt.cpp:39
  bool measure = false;  // measured added noise (hx_*_norms) instead of the bounds; power-of-two m
  // measured norms read back lazily (LazyLn above) instead of synchronously inside each operation
  static bool& deferNorms()
  {
    static bool on = false;
    return on;
  }
  bool lazy() const { return measure && deferNorms(); }

  const ChainContext* context;
  const Context* dev;
  const KeySet* keys;
  std::map<SKHandle, DoubleCRT> parts;
  PrimeSet primeSet;
  long ptxtSpace, intFactor = 1;
  LazyLn lnNoise;
  double ptxtMag = 1.0, lnRatFactor = 0.0;  // CKKS: |plaintext| bound, ln of the scaling factor
  // multiplyBy only: the tensor product's bookkeeping is done, its data not yet -- the four operand parts wait
  // here and `parts` holds storage for the three product parts.  The mod-switch that follows in reLinearize
  // (dropSmallAndSpecialPrimes) consumes them through hx_tensor_bring_to_set: the products are formed inside the
  // mod-down kernels; anything else that needs the data first calls materializeTensor().  Never visible outside
  // multiplyBy.
  struct PendingTensor {
    DoubleCRT c0, c1, d0, d1;
  };
  std::shared_ptr<PendingTensor> pendingTensor;
  void materializeTensor()
  {
    if (!pendingTensor)
      return;
    std::shared_ptr<PendingTensor> t = std::move(pendingTensor);
    pendingTensor.reset();
    helib_amd::tensorProduct(t->c0, t->c1, t->d0, t->d1, parts.at(SKHandle{0, 1}), parts.at(SKHandle{1, 1}),
                             parts.at(SKHandle{2, 1}));
  }

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
    materializeTensor();
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
    const double addedBound = modSwitchAddedNoiseBound(), logdiff = context->logOfProduct(diff);
    std::function<double()> added = modDownParts(one, inter, PrimeSet())[0];
    lnNoise.apply(dev, lazy(), [=](double& ln) {
      const double a = added();
      HELIB_AMD_STATS_UPDATE("mod-switch-added-noise", a / addedBound);   // src/Ctxt.cpp:535-537
      ln = detail::logaddexp(ln - logdiff, detail::ln(a));
    });
    lnRatFactor -= logdiff;  // ratFactor /= f (:533, :553)
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
      materializeTensor();
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
    materializeTensor();
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
  void multLowLvl(Ctxt other, bool lazyTensor = false)
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
    tensorProduct(other, lazyTensor);
  }
  void multiplyBy(const Ctxt& other)
  {
    HELIB_AMD_TIMER_START;
    multLowLvl(other, true);  // works on a copy of `other`, as the reference does (src/Ctxt.cpp:1716-1745)
    reLinearize();
    materializeTensor();
  }
  void multiplyBy(Ctxt&& other)  // the operand may be consumed: no copy
  {
    HELIB_AMD_TIMER_START;
    multLowLvl(std::move(other), true);
    reLinearize();
    materializeTensor();
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
    if (n_other > 1) {
      materializeTensor();
      reLinearizeMany();
      return;
    }
    const KeySwitch* W = keys->matrixFor(hnd);
    if (!W)
      throw LogicError("no key-switching matrices for this part");
    dropSmallAndSpecialPrimes();
    relin_CKKS_adjust();
    const IndexSet& sp = context->specialPrimes;
    // No mod-switch consumed the pending tensor product (a fresh CKKS product, a product at a level that needs
    // none): at the full level the key switch takes the operands themselves -- hx_mul_relin forms the product
    // parts inside the inverse transform's load and the key-switch kernel; otherwise the product is formed now.
    std::shared_ptr<PendingTensor> pend;
    if (pendingTensor && pendingTensor->c0.getIndexSet() == context->ctxtPrimes && toSet(context->ctxtPrimes) == primeSet &&
        W->coversInOrder(context->ctxtPrimes, sp)) {
      pend = std::move(pendingTensor);
      pendingTensor.reset();
    } else {
      materializeTensor();
    }
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
    const int batch = t0.batch();
    std::shared_ptr<std::vector<double>> nrm;
    if (pend) {
      if (measure) {
        dev->deferNorms(lazy());
        nrm = dev->normBuffer(digits.size() * (size_t)batch);
        check(hx_mul_relin_norms(pend->c0.handle(), pend->c1.handle(), pend->d0.handle(), pend->d1.handle(), W->handle(),
                                 idx.data(), off.data(), (int)digits.size(), o0.handle(), o1.handle(), nrm->data()));
      } else {
        check(hx_mul_relin(pend->c0.handle(), pend->c1.handle(), pend->d0.handle(), pend->d1.handle(), W->handle(),
                           idx.data(), off.data(), (int)digits.size(), o0.handle(), o1.handle()));
      }
    } else if (measure) {
      dev->deferNorms(lazy());
      nrm = dev->normBuffer(digits.size() * (size_t)batch);
      check(hx_relinearize_norms(t0.handle(), its == parts.end() ? nullptr : its->second.handle(), t2.handle(),
                                 W->handle(), idx.data(), off.data(), (int)digits.size(), sp.data(), (int)sp.size(),
                                 o0.handle(), o1.handle(), nrm->data()));
    } else {
      check(hx_relinearize(t0.handle(), its == parts.end() ? nullptr : its->second.handle(), t2.handle(),
                           W->handle(), idx.data(), off.data(), (int)digits.size(), sp.data(), (int)sp.size(),
                           o0.handle(), o1.handle()));
    }
    // noise: scaled parts + key-switch added noise (src/Ctxt.cpp:746, 827-841)
    std::vector<double> digitLn;
    for (auto& d : digits)
      digitLn.push_back(context->logOfProduct(d));
    const double uniform = std::log(context->noiseBoundForUniform(0.5, context->phim)), keyLn = keys->lnNoise;
    lnNoise.apply(dev, lazy(), [=](double& ln) {
      ln = detail::logaddexp(ln + logProd, keySwitchAddedNoise(nrm.get(), digitLn, batch, uniform, keyLn));
    });
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

  // ---- size / correctness accessors (include/helib/Ctxt.h:1291-1325, src/Ctxt.cpp:116-127) ----
  // ln of totalNoiseBound(): for CKKS ptxtMag*ratFactor + noiseBound, else noiseBound
  double lnTotalNoiseBound() const
  {
    const double ln = lnNoise;
    return context->ckks ? detail::logaddexp(detail::ln(ptxtMag) + lnRatFactor, ln) : ln;
  }
  // log2 of the modulus over the total noise bound
  double capacity() const { return (logOfPrimeSet() - std::max(lnTotalNoiseBound(), 0.0)) / std::log(2.0); }
  long bitCapacity() const { return (long)capacity(); }
  // totalNoiseBound * polyNormBnd <= 0.48 Q: would this ciphertext decrypt without errors?
  bool isCorrect() const;

  // ---- products of more than two ciphertexts ----
  // Ctxt::multiplyBy2 (src/Ctxt.cpp:1776-1828): the product of three ciphertexts with ONE
  // relinearisation at the end (parts up to s^3), multiplying in the order of their capacities
  void multiplyBy2(const Ctxt& other1, const Ctxt& other2)
  {
    HELIB_AMD_TIMER_START;
    if (parts.empty())
      return;
    if (other1.parts.empty()) {
      *this = other1;
      return;
    }
    if (other2.parts.empty()) {
      *this = other2;
      return;
    }
    const double cap = capacity(), cap1 = other1.capacity(), cap2 = other2.capacity();
    if (cap < cap1 && cap < cap2) {
      Ctxt tmp = other1;
      tmp.multLowLvl(other2);
      multLowLvl(std::move(tmp));
      reLinearize();
      return;
    }
    const bool swap = cap < cap2 || cap1 < cap2;
    Ctxt first = swap ? other2 : other1, second = swap ? other1 : other2;   // (copies: either may be *this)
    multLowLvl(std::move(first));
    multLowLvl(std::move(second));
    reLinearize();
  }
  void square() { multiplyBy(Ctxt(*this)); }                 // Ctxt::square
  void cube() { multiplyBy2(Ctxt(*this), Ctxt(*this)); }     // Ctxt::cube
  // Ctxt::power (src/polyEval.cpp:392-414): repeated squaring for a power of two, otherwise
  // DynamicCtxtPowers (:18-29): X^e = X^(e-k) * X^k, k the largest power of two below e
  void power(long e)
  {
    if (e < 1)
      throw InvalidArgument("Cannot raise a ctxt to a non positive exponent");
    if (e == 1)
      return;
    if ((e & (e - 1)) == 0) {
      for (; e > 1; e >>= 1)
        square();
      return;
    }
    std::map<long, Ctxt> powers;
    powers.emplace(1, *this);
    *this = powerRec(powers, e);
  }
  // Ctxt::frobeniusAutomorph (src/Ctxt.cpp:2526-2545): X -> X^(p^j) for BGV (j mod ord(p)); for CKKS
  // complex conjugation when j is odd
  void frobeniusAutomorph(long j)
  {
    if (parts.empty() || j == 0)
      return;
    if (context->ckks) {
      if (j & 1)
        smartAutomorph(context->m - 1);
      return;
    }
    const long m = context->m, p0 = ((context->p % m) + m) % m;
    long d = 1, x = p0;
    while (x != 1) {
      x = (long)((unsigned __int128)x * (unsigned long)p0 % (unsigned long)m);
      d++;
    }
    j = ((j % d) + d) % d;
    if (j)
      smartAutomorph((long)detail::powmod((uint64_t)p0, (uint64_t)j, (uint64_t)m));
  }

  // ---- plaintext constants ----
  // Ctxt::multByConstant(const DoubleCRT&, double size) (src/Ctxt.cpp:1832-1856), BGV: every part
  // times the constant (dcrt may live on more primes); size < 0: the bound for coefficients uniform
  // in [-ptxtSpace/2, ptxtSpace/2]
  void multByConstant(const DoubleCRT& dcrt, double size = -1.0)
  {
    if (parts.empty())
      return;
    if (size < 0.0)
      size = context->noiseBoundForMod(ptxtSpace, context->phim);
    for (auto& kv : parts)
      kv.second *= dcrt;
    lnNoise += detail::ln(size);
  }
  // Ctxt::multByConstant(const ZZ& / long) for BGV (src/Ctxt.cpp:2033-2110): c mod ptxtSpace = c1*d,
  // d = gcd(c, ptxtSpace); the ciphertext is multiplied by the balanced d only, the unit c1 goes
  // into intFactor (its inverse)
  void multByConstant(long c)
  {
    if (parts.empty())
      return;
    if (context->ckks) {
      multByConstantCKKS((double)c);
      return;
    }
    const long P = ptxtSpace;
    long c0 = ((c % P) + P) % P;
    if (c0 == 1)
      return;
    if (c0 == 0) {
      parts.clear();
      return;
    }
    const long d = std::gcd(c0, P);
    intFactor = (long)detail::mulmod((uint64_t)intFactor, (uint64_t)invMod(c0 / d, P), (uint64_t)P);
    if (d == 1)
      return;
    const long cc = d > P / 2 ? d - P : d;
    lnNoise += std::log((double)std::labs(cc));
    for (auto& kv : parts)
      kv.second *= cc;
  }
  // CKKS, scalar: no polynomial work at all -- ptxtMag *= |c|, ratFactor /= |c|, a sign flips the parts
  void multByConstantCKKS(double c)
  {
    if (parts.empty() || c == 1.0)
      return;
    if (c == 0.0) {
      parts.clear();
      return;
    }
    ptxtMag *= std::fabs(c);
    lnRatFactor -= std::log(std::fabs(c));
    if (c < 0)
      negate();
  }
  // Ctxt::multByConstantCKKS(const DoubleCRT&, size, factor, roundingErr) (src/Ctxt.cpp:1905-1938):
  // dcrt encodes a constant of magnitude <= size scaled by `factor` with encoding error <= roundingErr
  void multByConstantCKKS(const DoubleCRT& dcrt, double size, double factor, double roundingErr)
  {
    if (parts.empty())
      return;
    if (size <= 0)
      size = 1.0;
    if (factor <= 0 || roundingErr < 0)
      throw InvalidArgument("factor and roundingErr are the encoder's: pass them");
    const double n = lnNoise;
    lnNoise = detail::logaddexp(detail::logaddexp(n + std::log(factor) + std::log(size),
                                                  detail::ln(roundingErr) + lnRatFactor + detail::ln(ptxtMag)),
                                n + detail::ln(roundingErr));
    ptxtMag *= size;
    lnRatFactor += std::log(factor);
    for (auto& kv : parts)
      kv.second *= dcrt;
  }
  // Ctxt::addConstant(const DoubleCRT&, double size) (src/Ctxt.cpp:896-935), BGV: the constant is
  // scaled by f = balRem(intFactor * Q mod ptxtSpace) and added to the part of 1
  void addConstant(const DoubleCRT& dcrt, double size = -1.0)
  {
    if (size < 0.0)
      size = context->noiseBoundForMod(ptxtSpace, context->phim);
    long f = 1;
    if (ptxtSpace > 2) {
      const long p = ptxtSpace;
      f = (long)detail::mulmod(context->productOfPrimesMod(primeSet, (uint64_t)p), (uint64_t)(intFactor % p), (uint64_t)p);
      if (f > p / 2)
        f -= p;
    }
    lnNoise = detail::logaddexp(lnNoise, detail::ln(size * (double)std::labs(f)));
    addToPartOne(dcrt, f);
  }
  // Ctxt::addConstantCKKS(const DoubleCRT&, size, factor) (src/Ctxt.cpp:951-1045): the constant (scaled
  // by `factor`) is multiplied by round(ratFactor / factor) and added to the part of 1.  The reference
  // mod-switches up (addSomePrimes) when the rounded ratio is off by more than 2^-precision; here that
  // case is an error.
  void addConstantCKKS(const DoubleCRT& dcrt, double size, double factor)
  {
    if (size <= 0)
      size = 1.0;
    if (factor <= 0)
      throw InvalidArgument("factor is the encoder's: pass it");
    const double x = std::exp(lnRatFactor - std::log(factor));
    const double ratio = std::floor(x + 0.5);
    if (ratio < 1 || std::fabs(ratio / x - 1.0) * std::ldexp(1.0, (int)context->r) > 1.0)
      throw RuntimeError("addConstantCKKS: ratFactor / factor is too far from an integer "
                         "(the reference would call addSomePrimes here)");
    ptxtMag += size;
    lnNoise = detail::logaddexp(lnNoise, std::log(0.5));
    addToPartOne(dcrt, (long)ratio);
  }
  Ctxt& operator+=(const Ctxt& o)
  {
    addCtxt(o);
    return *this;
  }
  Ctxt& operator-=(const Ctxt& o)
  {
    addCtxt(o, true);
    return *this;
  }
  Ctxt& operator*=(const Ctxt& o)
  {
    multiplyBy(o);
    return *this;
  }

private:
  static long invMod(long v, long P)
  {
    long aa = ((v % P) + P) % P, bb = P, x0 = 1, x1 = 0;
    while (bb) {
      long q = aa / bb, t = aa % bb;
      aa = bb, bb = t;
      t = x0 - q * x1, x0 = x1, x1 = t;
    }
    if (aa != 1)
      throw LogicError("not invertible modulo the plaintext space");
    return ((x0 % P) + P) % P;
  }
  void addToPartOne(const DoubleCRT& dcrt, long f)
  {
    auto it = parts.find(SKHandle{0, 1});
    if (it == parts.end())
      throw RuntimeError("Ctxt::addPart: no part pointing at 1");
    if (f == 1) {
      it->second += dcrt;
    } else {
      DoubleCRT tmp = dcrt;
      tmp *= f;
      it->second += tmp;
    }
  }
  Ctxt powerRec(std::map<long, Ctxt>& powers, long n)
  {
    auto it = powers.find(n);
    if (it != powers.end())
      return it->second;
    long k = 1;
    while (2 * k < n)
      k *= 2;                                     // the largest power of two below n
    Ctxt c = powerRec(powers, n - k);
    c.multiplyBy(powerRec(powers, k));
    powers.emplace(n, c);
    return c;
  }
  // Ctxt::reLinearize with several non-canonical parts (after multiplyBy2: s^2 and s^3): parts 1 and s
  // are scaled by P, every other part goes through keySwitchPart -- break into digits, key-switch with
  // its own matrix, accumulate (src/Ctxt.cpp:720-842)
  void reLinearizeMany()
  {
    std::vector<std::pair<SKHandle, const KeySwitch*>> mats;
    for (auto& kv : parts)
      if (!kv.first.isOne() && !kv.first.isBase()) {
        const KeySwitch* W = keys->matrixFor(kv.first);
        if (!W)
          throw LogicError("no key-switching matrices for s^" + std::to_string(kv.first.powerOfS) + "(X^" +
                           std::to_string(kv.first.powerOfX) + ")");
        mats.emplace_back(kv.first, W);
      }
    dropSmallAndSpecialPrimes();
    relin_CKKS_adjust();
    materializeTensor();
    const IndexSet& sp = context->specialPrimes;
    const double logProd = context->logOfProduct(sp);
    lnRatFactor += logProd;
    std::vector<IndexSet> digits;
    for (auto& d : context->digits) {
      IndexSet r;
      for (int i : d)
        if (primeSet.count(i))
          r.push_back(i);
      if (!r.empty())
        digits.push_back(r);
    }
    if (ptxtSpace > 1) {
      ptxtSpace = std::gcd(ptxtSpace, keys->ptxtSpace ? keys->ptxtSpace : context->ptxtSpace);
      intFactor %= ptxtSpace;
    }
    DoubleCRT part0 = std::move(parts.at(SKHandle{0, 1}));
    part0.addPrimesAndScale(sp);
    auto its = parts.find(SKHandle{1, 1});
    DoubleCRT part1 = its == parts.end() ? DoubleCRT(*dev, part0.getIndexSet(), part0.batch()) : std::move(its->second);
    if (its != parts.end())
      part1.addPrimesAndScale(sp);
    const int batch = part0.batch();
    std::vector<std::shared_ptr<std::vector<double>>> norms;   // per key-switched part (measured)
    for (auto& hw : mats) {
      std::shared_ptr<std::vector<double>> nrm;
      if (measure) {
        dev->deferNorms(lazy());
        nrm = dev->normBuffer(digits.size() * (size_t)batch);
      }
      DoubleCRT dg = parts.at(hw.first).breakIntoDigits(digits, sp, nrm ? nrm->data() : nullptr);
      keySwitchDigits(*hw.second, dg, part0, part1);
      norms.push_back(nrm);
    }
    std::vector<double> digitLn;
    for (auto& d : digits)
      digitLn.push_back(context->logOfProduct(d));
    const double uniform = std::log(context->noiseBoundForUniform(0.5, context->phim)), keyLn = keys->lnNoise;
    lnNoise.apply(dev, lazy(), [=](double& ln) {
      double added = -INFINITY;
      for (auto& nrm : norms)
        added = detail::logaddexp(added, keySwitchAddedNoise(nrm.get(), digitLn, batch, uniform, keyLn));
      ln = detail::logaddexp(ln + logProd, added);
    });
    parts.clear();
    parts.emplace(SKHandle{0, 1}, std::move(part0));
    parts.emplace(SKHandle{1, 1}, std::move(part1));
    primeSet = primeSet | toSet(sp);
  }
  static bool canonicalPair(const std::map<SKHandle, DoubleCRT>& ps)
  {
    return ps.size() == 2 && ps.count(SKHandle{0, 1}) && ps.count(SKHandle{1, 1});
  }
  void tensorProduct(const Ctxt& o, bool lazy = false)
  {
    if (ptxtSpace > 2) {
      uint64_t q = context->productOfPrimesMod(primeSet, (uint64_t)ptxtSpace);
      intFactor = (long)detail::mulmod(detail::mulmod((uint64_t)intFactor, (uint64_t)o.intFactor, (uint64_t)ptxtSpace), q,
                                       (uint64_t)ptxtSpace);
    }
    if (canonicalPair(parts) && canonicalPair(o.parts)) {
      const DoubleCRT &c0 = parts.at(SKHandle{0, 1}), &c1 = parts.at(SKHandle{1, 1});
      const DoubleCRT &d0 = o.parts.at(SKHandle{0, 1}), &d1 = o.parts.at(SKHandle{1, 1});
      IndexSet idx = c0.getIndexSet();
      DoubleCRT::Uninitialized u;
      DoubleCRT t0(*dev, idx, c0.batch(), u), t1(*dev, idx, c0.batch(), u), t2(*dev, idx, c0.batch(), u);
      static const bool lazy_off = std::getenv("HX_NO_LAZY_TENSOR") != nullptr;   // (A/B switch)
      if (lazy && !lazy_off)   // copies of the operand parts are copy-on-write handles: nothing moves
        pendingTensor = std::make_shared<PendingTensor>(PendingTensor{c0, c1, d0, d1});
      else
        helib_amd::tensorProduct(c0, c1, d0, d1, t0, t1, t2);
      parts.clear();
      parts.emplace(SKHandle{0, 1}, std::move(t0));
      parts.emplace(SKHandle{1, 1}, std::move(t1));
      parts.emplace(SKHandle{2, 1}, std::move(t2));
    } else {  // any parts (src/Ctxt.cpp:1576-1597): all pairwise products, accumulated by handle
      std::map<SKHandle, DoubleCRT> np;
      for (auto& a : parts)
        for (auto& b : o.parts) {
          SKHandle h;
          if (!h.mul(a.first, b.first))
            throw LogicError("cannot multiply parts under different automorphisms");
          DoubleCRT t = a.second;
          t *= b.second;
          auto it = np.find(h);
          if (it == np.end())
            np.emplace(h, std::move(t));
          else
            it->second += t;
        }
      parts = std::move(np);
    }
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
  // sum over the digits of (norm of the digit, measured or the uniform bound) * P_digit * the matrix' noise, as ln
  static double keySwitchAddedNoise(const std::vector<double>* nrm, const std::vector<double>& digitLn, int batch,
                                    double uniformLn, double keyLn)
  {
    double added = -INFINITY;
    for (size_t k = 0; k < digitLn.size(); k++) {
      double nb = uniformLn;
      if (nrm) {   // norm_val = embeddingLargestCoeff(digit) (src/DoubleCRT.cpp:538-545), largest of the batch
        double mx = 0;
        for (int b = 0; b < batch; b++)
          mx = std::max(mx, (*nrm)[k * (size_t)batch + (size_t)b]);
        nb = detail::ln(mx);
      }
      added = detail::logaddexp(added, nb + digitLn[k] + keyLn);
    }
    return added;
  }
  // polynomial work of modDownToSet (after a mod-up by `add`) on all parts of ciphertexts that share
  // one prime set: one fused call; returns, per ciphertext, a function giving its added noise -- measured: the
  // sum over parts of embeddingLargestCoeff(fdelta) * h^power (src/Ctxt.cpp:495-527), to be called once the norms
  // have been read back (LazyLn does); otherwise the bound
  static std::vector<std::function<double()>> modDownParts(std::vector<Ctxt*>& cts, const PrimeSet& keep,
                                                           const PrimeSet& add)
  {
    Ctxt& a = *cts[0];
    std::vector<hx_poly*> polys;
    for (Ctxt* c : cts)
      for (auto& kv : c->parts)
        polys.push_back(kv.second.handle());
    PrimeSet cur = a.primeSet | add;
    IndexSet drop = toVec(cur - keep), addv = toVec(add);
    const int batch = cts[0]->parts.begin()->second.batch();
    uint64_t pt = (uint64_t)a.ptxtSpace;
    std::shared_ptr<std::vector<double>> norms;
    int rc;
    std::shared_ptr<PendingTensor> pend;
    if (cts.size() == 1 && a.pendingTensor) {   // the tensor product is formed inside this mod-switch
      pend = std::move(a.pendingTensor);
      a.pendingTensor.reset();
    } else {
      for (Ctxt* c : cts)
        c->materializeTensor();
    }
    if (pend) {
      hx_poly* o0 = a.parts.at(SKHandle{0, 1}).handle();
      hx_poly* o1 = a.parts.at(SKHandle{1, 1}).handle();
      hx_poly* o2 = a.parts.at(SKHandle{2, 1}).handle();
      if (a.measure) {
        a.dev->deferNorms(a.lazy());
        norms = a.dev->normBuffer(3 * (size_t)batch);
        rc = hx_tensor_bring_to_set_norms(pend->c0.handle(), pend->c1.handle(), pend->d0.handle(), pend->d1.handle(), o0, o1,
                                          o2, addv.data(), (int)addv.size(), drop.data(), (int)drop.size(), pt, norms->data());
      } else {
        rc = hx_tensor_bring_to_set(pend->c0.handle(), pend->c1.handle(), pend->d0.handle(), pend->d1.handle(), o0, o1, o2,
                                    addv.data(), (int)addv.size(), drop.data(), (int)drop.size(), pt);
      }
    } else if (a.measure) {
      a.dev->deferNorms(a.lazy());
      norms = a.dev->normBuffer(polys.size() * (size_t)batch);
      rc = addv.empty() ? hx_scale_down_multi_norms(polys.data(), (int)polys.size(), drop.data(), (int)drop.size(), pt,
                                                    norms->data(), nullptr)
                        : hx_bring_to_set_multi_norms(polys.data(), (int)polys.size(), addv.data(), (int)addv.size(),
                                                      drop.data(), (int)drop.size(), pt, norms->data());
    } else {
      rc = addv.empty() ? hx_scale_down_multi(polys.data(), (int)polys.size(), drop.data(), (int)drop.size(), pt)
                        : hx_bring_to_set_multi(polys.data(), (int)polys.size(), addv.data(), (int)addv.size(),
                                                drop.data(), (int)drop.size(), pt);
    }
    if (rc != HX_OK && pend)
      a.pendingTensor = pend;   // the product was not formed: `parts` still holds no data, the operands wait again
    check(rc);
    std::vector<std::function<double()>> out;
    size_t k = 0;
    const double h = a.context->skBound();
    for (Ctxt* c : cts) {
      if (!a.measure) {
        const double bound = c->modSwitchAddedNoiseBound();
        out.push_back([bound] { return bound; });
        continue;
      }
      std::vector<double> weight;   // h^powerOfS per part, in the order the parts were listed
      for (auto& kv : c->parts)
        weight.push_back(std::pow(h, (double)kv.first.powerOfS));
      const size_t k0 = k;
      k += weight.size();
      out.push_back([norms, weight, k0, batch] {
        double sum = 0;
        for (size_t i = 0; i < weight.size(); i++) {
          double mx = 0;
          for (int b = 0; b < batch; b++)
            mx = std::max(mx, (*norms)[(k0 + i) * (size_t)batch + (size_t)b]);
          sum += mx * weight[i];
        }
        return sum;
      });
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
    std::vector<std::function<double()>> added;
    if (diff.empty()) {  // pure mod-up
      IndexSet d = toVec(add);
      for (Ctxt* c : cts)
        c->materializeTensor();
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
        const double logdiff = c->context->logOfProduct(diff);
        std::function<double()> a = added[i];
        c->lnNoise.apply(c->dev, c->lazy(), [=](double& ln) { ln = detail::logaddexp(ln - logdiff, detail::ln(a())); });
        c->primeSet = inter;
      }
    }
  }
};

inline std::pair<double, double> Ctxt::computeIntervalForMul(const Ctxt& c1, const Ctxt& c2)
{
  const double LN2 = std::log(2.0);
  double cap1 = c1.logOfPrimeSet() - std::max((double)c1.lnNoise, 0.0);
  double cap2 = c2.logOfPrimeSet() - std::max((double)c2.lnNoise, 0.0);
  double adn1 = std::log(c1.modSwitchAddedNoiseBound()), adn2 = std::log(c2.modSwitchAddedNoiseBound());
  if (c1.context->ckks) {  // the opposite end: keep n*q'/q above the added noise (:1637-1651)
    double lo = std::max(cap1 + adn1, cap2 + adn2) + safety;
    return {lo, lo + 4 * LN2};
  }
  double hi = std::min(cap1 + adn1, cap2 + adn2) - safety;
  return {hi - 4 * LN2, hi};
}

// Phi_m(X), coefficients a_0 .. a_phi(m) (monic): (X^m - 1) / prod_{d | m, d < m} Phi_d by exact division
// over the integers (the reference: Cyclotomic, src/NumbTh.cpp:1034-1060)
inline std::vector<long> cyclotomic(long m)
{
  std::map<long, std::vector<long>> phi;
  std::vector<long> divs;
  for (long d = 1; d <= m; d++)
    if (m % d == 0)
      divs.push_back(d);
  for (long d : divs) {
    std::vector<long> num((size_t)d + 1, 0);
    num[(size_t)d] = 1;
    num[0] = -1;
    for (long e : divs)
      if (e < d && d % e == 0) {
        const std::vector<long>& den = phi[e];
        const size_t dd = den.size() - 1;   // degree of the (monic) divisor
        std::vector<long> quo(num.size() - dd, 0);
        for (size_t i = num.size() - 1; i + 1 > dd; i--) {
          const long c = num[i];
          quo[i - dd] = c;
          if (c)
            for (size_t j = 0; j <= dd; j++)
              num[i - dd + j] -= c * den[j];
          if (i == 0)
            break;
        }
        num = quo;
      }
    phi[d] = num;
  }
  return phi[m];
}

// calcPolyNormBnd (src/PAlgebra.cpp:215-434): the ring constant c_M.  1 for a power of two,
// 2 cot(pi/(2u))/u when the odd part of m is a power of one prime u; otherwise, with m replaced by the
// radical of its odd part, the maximal absolute row sum of the inverse of the Vandermonde matrix of the
// primitive m-th roots x_j: row i of column j is q_i(x_j) / Phi_m'(x_j) with the Horner prefixes q_0 = 1,
// q_i = q_(i-1) x_j + a_(n-i) of Phi_m (:360-372); |Phi_m'(x_j)| = prod_i |x_i - x_j| as the exponential
// of a sum of logarithms.  O(phi(m)^2), computed once per m.
inline double polyNormBnd(long m)
{
  while (m % 2 == 0)
    m /= 2;
  if (m == 1)
    return 1.0;
  std::vector<long> fac;
  for (long r = m, d = 3; r > 1; d += 2)
    if (r % d == 0) {
      fac.push_back(d);
      while (r % d == 0)
        r /= d;
    }
  const double PI = std::acos(-1.0);
  if (fac.size() == 1)
    return 2.0 / std::tan(PI / (2.0 * fac[0])) / fac[0];
  m = 1;
  for (long u : fac)
    m *= u;
  const std::vector<long> a = cyclotomic(m);   // a_0 .. a_n, a_n = 1
  const size_t n = a.size() - 1;
  std::vector<long> res;
  for (long i = 1; i < m; i++)
    if (std::gcd(i, m) == 1)
      res.push_back(i);
  std::vector<double> logd((size_t)m, 0.0), cs((size_t)m), sn((size_t)m);
  for (long k = 0; k < m; k++) {
    cs[(size_t)k] = std::cos(2 * PI * (double)k / (double)m);
    sn[(size_t)k] = std::sin(2 * PI * (double)k / (double)m);
    if (k)
      logd[(size_t)k] = std::log(2.0 * std::sin(PI * (double)k / (double)m));
  }
  std::vector<double> inv_prod(n), qr(n, 1.0), qi(n, 0.0);
  for (size_t j = 0; j < n; j++) {
    double t = 0;
    for (size_t i = 0; i < n; i++)
      t += logd[(size_t)(((res[i] - res[j]) % m + m) % m)];
    inv_prod[j] = std::exp(-t);
  }
  double best = 0;
  for (size_t j = 0; j < n; j++)
    best += inv_prod[j];            // row 0: q_0 = 1
  for (size_t i = 1; i < n; i++) {
    double row = 0;
    for (size_t j = 0; j < n; j++) {
      const double xr = cs[(size_t)res[j]], xi = sn[(size_t)res[j]];
      const double nr = qr[j] * xr - qi[j] * xi + (double)a[n - i], ni = qr[j] * xi + qi[j] * xr;
      qr[j] = nr;
      qi[j] = ni;
      row += std::hypot(nr, ni) * inv_prod[j];
    }
    best = std::max(best, row);
  }
  return best;
}
inline bool Ctxt::isCorrect() const
{
  static std::map<long, double> cache;   // the ring constant, computed once per m
  static std::mutex guard;
  double cm;
  {
    std::lock_guard<std::mutex> lock(guard);
    auto it = cache.find(context->m);
    if (it == cache.end())
      it = cache.emplace(context->m, polyNormBnd(context->m)).first;
    cm = it->second;
  }
  return lnTotalNoiseBound() + std::log(cm) <= std::log(0.48) + logOfPrimeSet();
}

// ---- products of many ciphertexts (src/Ctxt.cpp:2803-2904) ----
namespace detail {
inline size_t splitBelow(size_t n)   // the highest power of two below n (n/2 <= n1 < n)
{
  size_t k = 1;
  while (2 * k < n)
    k *= 2;
  return k;
}
inline void incrementalProductRec(std::vector<Ctxt>& v, size_t lo, size_t n)
{
  if (n <= 1)
    return;
  const size_t n1 = splitBelow(n);
  incrementalProductRec(v, lo, n1);
  incrementalProductRec(v, lo + n1, n - n1);
  for (size_t i = lo + n1; i < lo + n; i++)
    v[i].multiplyBy(v[lo + n1 - 1]);
}
inline Ctxt totalProductRec(const std::vector<Ctxt>& v, size_t lo, size_t n)
{
  Ctxt out = v[lo];
  if (n == 2)
    out.multiplyBy(v[lo + 1]);
  else if (n == 3)
    out.multiplyBy2(v[lo + 1], v[lo + 2]);
  else if (n > 3) {
    const size_t n1 = splitBelow(n);
    out = totalProductRec(v, lo, n1);
    out.multiplyBy(totalProductRec(v, lo + n1, n - n1));
  }
  return out;
}
}  // namespace detail
// for i = n-1 .. 0: v[i] = prod_{j <= i} v[j], depth log n and (n log n)/2 products, in place
inline void incrementalProduct(std::vector<Ctxt>& v) { detail::incrementalProductRec(v, 0, v.size()); }
// prod_i v[i] in depth log n with n-1 products (triples through multiplyBy2)
inline Ctxt totalProduct(const std::vector<Ctxt>& v)
{
  if (v.empty())
    throw InvalidArgument("totalProduct of an empty vector");
  return detail::totalProductRec(v, 0, v.size());
}
// sum_i v1[i] * v2[i] with the low-level product and ONE relinearisation at the end
inline Ctxt innerProduct(const std::vector<Ctxt>& v1, const std::vector<Ctxt>& v2)
{
  const size_t n = std::min(v1.size(), v2.size());
  if (n == 0)
    throw InvalidArgument("innerProduct of empty vectors");
  Ctxt result = v1[0];
  result.multLowLvl(v2[0]);
  for (size_t i = 1; i < n; i++) {
    Ctxt tmp = v1[i];
    tmp.multLowLvl(v2[i]);
    result.addCtxt(tmp);
  }
  result.reLinearize();
  return result;
}

// Hoisting (src/matmul.cpp:48-184): break the `s` part of a ciphertext into digits ONCE, then every
// automorphism rotates the digits (a permutation of evaluation rows, hx_automorph on the whole digit
// block) and key-switches them with the matrix of that automorphism -- no inverse transform, no basis
// extension and no forward transforms per rotation.
class BasicAutomorphPrecon {
public:
  explicit BasicAutomorphPrecon(const Ctxt& ct) : ctxt_(ct)
  {
    Ctxt& c = ctxt_;
    if (c.parts.size() <= 1)
      return;
    c.cleanUp();
    if (c.parts.size() != 2 || !c.parts.count(SKHandle{0, 1}) || !c.parts.count(SKHandle{1, 1}))
      throw LogicError("Ciphertext is not in canonical form");
    const ChainContext& cc = *c.context;
    const IndexSet& sp = cc.specialPrimes;
    for (auto& d : cc.digits) {
      IndexSet r;
      for (int i : d)
        if (c.primeSet.count(i))
          r.push_back(i);
      if (!r.empty())
        digits_.push_back(r);
    }
    const DoubleCRT& ps = c.parts.at(SKHandle{1, 1});
    std::vector<double> nrm(c.measure ? digits_.size() * (size_t)ps.batch() : 0);
    if (c.measure)
      c.dev->deferNorms(false);   // (this estimate is needed right away: the numbers before the call returns)
    polyDigits_ = std::make_unique<DoubleCRT>(ps.breakIntoDigits(digits_, sp, c.measure ? nrm.data() : nullptr));
    // addedNoise = breakIntoDigits' return value * the matrices' noise bound (src/matmul.cpp:91-97);
    // noise = ctxt.noise * P + addedNoise (:99-112)
    double added = -INFINITY;
    for (size_t k = 0; k < digits_.size(); k++) {
      double nb;
      if (c.measure) {
        double mx = 0;
        for (int b = 0; b < ps.batch(); b++)
          mx = std::max(mx, nrm[k * (size_t)ps.batch() + (size_t)b]);
        nb = detail::ln(mx);
      } else {
        nb = std::log(cc.noiseBoundForUniform(0.5, cc.phim));
      }
      added = detail::logaddexp(added, nb + cc.logOfProduct(digits_[k]));
    }
    added += c.keys->lnNoise;
    lnNoise_ = detail::logaddexp(c.lnNoise + cc.logOfProduct(sp), added);
  }

  // the ciphertext rotated by X -> X^k
  Ctxt automorph(long k) const
  {
    const Ctxt& c = ctxt_;
    const ChainContext& cc = *c.context;
    const long m = cc.m;
    k = ((k % m) + m) % m;
    if (k == 1 || c.parts.empty())
      return c;
    if (std::gcd(k, m) != 1)
      throw InvalidArgument("k must be in Zm*");
    const IndexSet& sp = cc.specialPrimes;
    Ctxt res(cc, *c.dev, *c.keys);
    res.measure = c.measure;
    res.ptxtSpace = c.ptxtSpace;
    res.intFactor = c.intFactor;
    res.ptxtMag = c.ptxtMag;
    res.lnRatFactor = c.lnRatFactor + cc.logOfProduct(sp);
    res.primeSet = c.primeSet | toSet(sp);
    const DoubleCRT& p0 = c.parts.at(SKHandle{0, 1});
    if (c.parts.size() == 1) {   // only the constant part: nothing to key-switch (:145-151)
      DoubleCRT part0 = p0;
      part0.automorph(k);
      part0.addPrimesAndScale(sp);
      res.parts.emplace(SKHandle{0, 1}, std::move(part0));
      res.lnNoise = c.lnNoise + cc.logOfProduct(sp);
      return res;
    }
    const long amt = c.keys->firstStep(k);   // first key-switching matrix on the way to k (:153-162)
    auto it = amt ? c.keys->automorph.find(amt) : c.keys->automorph.end();
    if (it == c.keys->automorph.end())
      throw LogicError("no key-switching matrices for k=" + std::to_string(k));
    DoubleCRT part0 = p0;
    part0.automorph(amt);
    part0.addPrimesAndScale(sp);
    DoubleCRT dg = *polyDigits_;
    dg.automorph(amt);
    DoubleCRT part1(*c.dev, part0.getIndexSet(), part0.batch());
    keySwitchDigits(*it->second, dg, part0, part1);
    res.parts.emplace(SKHandle{0, 1}, std::move(part0));
    res.parts.emplace(SKHandle{1, 1}, std::move(part1));
    res.lnNoise = lnNoise_;
    if (amt != k) {   // more automorphisms to do: the usual smartAutomorph (:177-181)
      long inv = 1, a = amt % m, e = ChainContext::eulerPhi(m) - 1;
      while (e) {
        if (e & 1)
          inv = (long)((unsigned __int128)inv * (unsigned long)a % (unsigned long)m);
        a = (long)((unsigned __int128)a * (unsigned long)a % (unsigned long)m);
        e >>= 1;
      }
      res.smartAutomorph((long)((unsigned __int128)k * (unsigned long)inv % (unsigned long)m));
    }
    return res;
  }

private:
  Ctxt ctxt_;
  std::unique_ptr<DoubleCRT> polyDigits_;
  std::vector<IndexSet> digits_;
  double lnNoise_ = 0;
};

}  // namespace helib_amd

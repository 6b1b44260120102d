// helib_amd.hpp -- C++17 host facade over the C ABI (helib_amd.h) with the reference's names,
// argument meaning and error behaviour for this path, so that code written against
// include/helib/DoubleCRT.h / CModulus.h / keySwitching.h reads the same:
//
//   helib_amd::Context      Context::moduli + zMStar      (include/helib/Context.h:117, 339-366)
//   helib_amd::DoubleCRT    DoubleCRT                     (include/helib/DoubleCRT.h:212-385)
//   helib_amd::KeySwitch    KeySwitch (b columns + expanded a columns, keySwitching.h:86-101)
//   helib_amd::Cmodulus     Cmodulus: FFT / iFFT of one row (include/helib/CModulus.h:56-145)
//   helib_amd::BigInt       the ZZ coefficients of DoubleCRT::toPoly (src/DoubleCRT.cpp:925-1113)
//   tensorProduct / keySwitchDigits / multiplyBy / reLinearize   (src/Ctxt.cpp:191-230, 720-842, 1563-1774)
//
// Exceptions mirror include/helib/exceptions.h: InvalidArgument for bad arguments, RuntimeError
// for index-set mismatches / k not in Zm*, LogicError otherwise; messages come from the library
// (hx_last_error) and are the reference's own where one exists.
// Header-only; link with -lhelib_amd.  No NTL: coefficients cross the boundary as uint64 rows.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "helib_amd.h"

namespace helib_amd {

struct RuntimeError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct LogicError : std::logic_error {
  using std::logic_error::logic_error;
};
struct InvalidArgument : std::invalid_argument {
  using std::invalid_argument::invalid_argument;
};

inline void check(int rc)
{
  if (rc == HX_OK)
    return;
  std::string msg = hx_last_error();
  switch (rc) {
    case HX_ERR_INVALID: throw InvalidArgument(msg);
    case HX_ERR_PRIMESET:
    case HX_ERR_NOT_IN_ZMSTAR:
    case HX_ERR_DEVICE:
    case HX_ERR_NOMEM: throw RuntimeError(msg);
    default: throw LogicError(msg);
  }
}

using IndexSet = std::vector<int>;  // prime indices, Context::moduli order

// A signed big integer for the coefficients DoubleCRT::toPoly returns (the reference hands out NTL::ZZ,
// src/DoubleCRT.cpp:925-1113; NTL and GMP are not dependencies of this host).  Sign and magnitude,
// 64-bit limbs, little endian, no leading zero limbs; just what the CRT reconstruction and its
// callers' checks need.  toString() is decimal; limbs()/negative() feed mpz_import where GMP is at hand.
class BigInt {
public:
  BigInt() = default;
  explicit BigInt(uint64_t v)
  {
    if (v)
      mag_.push_back(v);
  }
  bool isZero() const { return mag_.empty(); }
  bool negative() const { return neg_; }
  const std::vector<uint64_t>& limbs() const { return mag_; }
  // this = this * m + a   (non-negative values)
  void mulAdd(uint64_t m, uint64_t a)
  {
    unsigned __int128 carry = a;
    for (auto& w : mag_) {
      unsigned __int128 t = (unsigned __int128)w * m + carry;
      w = (uint64_t)t;
      carry = t >> 64;
    }
    if (carry)
      mag_.push_back((uint64_t)carry);
  }
  // magnitude comparison: -1, 0, 1
  static int cmpMag(const BigInt& a, const BigInt& b)
  {
    if (a.mag_.size() != b.mag_.size())
      return a.mag_.size() < b.mag_.size() ? -1 : 1;
    for (size_t i = a.mag_.size(); i-- > 0;)
      if (a.mag_[i] != b.mag_[i])
        return a.mag_[i] < b.mag_[i] ? -1 : 1;
    return 0;
  }
  // |big| - |small| for |big| >= |small|
  static BigInt subMag(const BigInt& big, const BigInt& small)
  {
    BigInt r;
    r.mag_ = big.mag_;
    uint64_t borrow = 0;
    for (size_t i = 0; i < r.mag_.size(); i++) {
      const uint64_t s = i < small.mag_.size() ? small.mag_[i] : 0;
      const uint64_t a = r.mag_[i];
      const uint64_t d = a - s - borrow;
      borrow = (a < s || (a == s && borrow)) ? 1 : 0;
      r.mag_[i] = d;
    }
    while (!r.mag_.empty() && r.mag_.back() == 0)
      r.mag_.pop_back();
    return r;
  }
  BigInt negated() const
  {
    BigInt r = *this;
    r.neg_ = !mag_.empty() && !neg_;
    return r;
  }
  // (this + 1) / 2 for a non-negative value: the reference's prod_half
  BigInt halfUp() const
  {
    BigInt r = *this;
    uint64_t carry = 1;
    for (auto& w : r.mag_) {
      w += carry;
      carry = (w == 0 && carry) ? 1 : 0;
      if (!carry)
        break;
    }
    if (carry)
      r.mag_.push_back(1);
    uint64_t hi = 0;
    for (size_t i = r.mag_.size(); i-- > 0;) {
      const uint64_t w = r.mag_[i];
      r.mag_[i] = (w >> 1) | (hi << 63);
      hi = w & 1;
    }
    while (!r.mag_.empty() && r.mag_.back() == 0)
      r.mag_.pop_back();
    return r;
  }
  // the value modulo q, in [0, q)  (mathematical residue: a negative value gives q - (|v| mod q))
  uint64_t mod(uint64_t q) const
  {
    unsigned __int128 r = 0;
    for (size_t i = mag_.size(); i-- > 0;)
      r = ((r << 64) | mag_[i]) % q;
    uint64_t v = (uint64_t)r;
    return (neg_ && v) ? q - v : v;
  }
  double toDouble() const
  {
    double v = 0;
    for (size_t i = mag_.size(); i-- > 0;)
      v = v * 18446744073709551616.0 + (double)mag_[i];
    return neg_ ? -v : v;
  }
  std::string toString() const
  {
    if (mag_.empty())
      return "0";
    std::vector<uint64_t> t = mag_;
    std::string digits;
    while (!t.empty()) {
      unsigned __int128 r = 0;
      for (size_t i = t.size(); i-- > 0;) {
        unsigned __int128 cur = (r << 64) | t[i];
        t[i] = (uint64_t)(cur / 10000000000000000000ull);
        r = cur % 10000000000000000000ull;
      }
      while (!t.empty() && t.back() == 0)
        t.pop_back();
      uint64_t chunk = (uint64_t)r;
      for (int i = 0; i < 19; i++) {
        digits.push_back((char)('0' + chunk % 10));
        chunk /= 10;
        if (t.empty() && chunk == 0)
          break;
      }
    }
    if (neg_)
      digits.push_back('-');
    return std::string(digits.rbegin(), digits.rend());
  }

private:
  bool neg_ = false;
  std::vector<uint64_t> mag_;
};

class Context {
public:
  explicit Context(uint64_t m, int device = 0) : m_(m)
  {
    hx_ctx* c = nullptr;
    check(hx_ctx_create(&c, device, m));
    h_.reset(c, [](hx_ctx* p) { hx_ctx_destroy(p); });
    uint64_t n = 0;
    check(hx_ctx_phim(c, &n));
    phim_ = (long)n;
  }
  // Cmodulus(zms, q, root): root = NTL's RootTable[0][k] for m = 2^k, FindPrimitiveRoot output
  // otherwise; 0 = FindPrimRootT.  Returns the index in Context::moduli.
  long addPrime(uint64_t q, uint64_t root = 0)
  {
    int idx = -1;
    check(hx_ctx_add_prime(h_.get(), q, root, &idx));
    uint64_t qq, rr;
    check(hx_ctx_prime(h_.get(), idx, &qq, &rr));
    primes_.push_back(qq);
    roots_.push_back(rr);
    return idx;
  }
  long getM() const { return (long)m_; }
  long getPhiM() const { return phim_; }
  long ithPrime(long i) const { return (long)primes_.at((size_t)i); }
  uint64_t ithRoot(long i) const { return roots_.at((size_t)i); }
  long numPrimes() const { return (long)primes_.size(); }
  void sync() const { check(hx_ctx_sync(h_.get())); }
  // the HIP stream (hipStream_t) every call on this context enqueues on; nullptr = the default stream
  void setStream(void* stream) const { check(hx_ctx_set_stream(h_.get(), stream)); }
  // device memory for the slabs of this context's DoubleCRT objects, reserved up front (hx_ctx_reserve)
  void reserve(uint64_t bytes) const { check(hx_ctx_reserve(h_.get(), bytes)); }
  // reserved bytes, bytes in use, hipMalloc calls so far, blocks parked for a live graph (hx_ctx_arena_stats)
  void arenaStats(uint64_t out[4]) const { check(hx_ctx_arena_stats(h_.get(), out)); }
  hx_ctx* handle() const { return h_.get(); }

  // Measured-noise norms (hx_*_norms) either land in the caller's array before the call returns, or --
  // deferNorms(true), hx_ctx_defer_norms -- when the context is next flushed, so that the host can keep
  // enqueueing work while the norm kernels run.  A deferred array must outlive the flush whatever becomes
  // of the ciphertext that asked for it: normBuffer() hands out arrays this object keeps until then.
  void deferNorms(bool on) const
  {
    std::lock_guard<std::mutex> lock(norms_->mu);
    if (on != norms_->defer) {
      check(hx_ctx_defer_norms(h_.get(), on ? 1 : 0));   // switching off flushes
      norms_->defer = on;
      if (!on)
        norms_->kept.clear();
    }
  }
  bool deferringNorms() const { return norms_->defer; }
  std::shared_ptr<std::vector<double>> normBuffer(size_t n) const
  {
    auto buf = std::make_shared<std::vector<double>>(n, 0.0);
    std::lock_guard<std::mutex> lock(norms_->mu);
    if (norms_->defer) {
      if (norms_->kept.size() >= 256) {   // nobody is reading: complete them, keep the list short
        check(hx_norms_flush(h_.get()));
        norms_->kept.clear();
      }
      norms_->kept.push_back(buf);
    }
    return buf;
  }
  void flushNorms() const
  {
    std::lock_guard<std::mutex> lock(norms_->mu);
    check(hx_norms_flush(h_.get()));
    norms_->kept.clear();
  }
  size_t pendingNormBuffers() const { return norms_->kept.size(); }

  // HIP graphs (helib_amd.h: hx_ctx_graph_begin / _end): everything enqueued on this context between
  // graphBegin() and graphEnd() is recorded instead of run; Graph::launch() replays it with one launch
  // on the same buffers.  For the launch-bound case: one ciphertext at a time.
  class Graph {
  public:
    explicit Graph(hx_graph* g) : g_(g, [](hx_graph* p) { hx_graph_destroy(p); }) {}
    void launch() const { check(hx_graph_launch(g_.get())); }

  private:
    std::shared_ptr<hx_graph> g_;
  };
  void graphBegin() const { check(hx_ctx_graph_begin(h_.get())); }
  Graph graphEnd() const
  {
    hx_graph* g = nullptr;
    check(hx_ctx_graph_end(h_.get(), &g));
    return Graph(g);
  }

private:
  uint64_t m_;
  long phim_ = 0;
  std::shared_ptr<hx_ctx> h_;
  std::vector<uint64_t> primes_, roots_;
  // copies of a Context share the device context, so they share its deferral state too: the arrays a
  // deferred read-back will write stay alive until the flush whichever copy asked for them
  struct NormState {
    std::mutex mu;
    bool defer = false;
    std::vector<std::shared_ptr<std::vector<double>>> kept;
  };
  std::shared_ptr<NormState> norms_ = std::make_shared<NormState>();
};

class DoubleCRT {
public:
  // DoubleCRT(context, indexSet): all-zero rows; `batch` independent objects share the prime set
  DoubleCRT(const Context& context, const IndexSet& s, int batch = 1) : context_(&context), batch_(batch)
  {
    hx_poly* p = nullptr;
    check(hx_poly_create(context.handle(), batch, s.data(), (int)s.size(), &p));
    h_.reset(p);
  }
  // storage only, contents unspecified (an output about to be overwritten by the engine)
  struct Uninitialized {};
  DoubleCRT(const Context& context, const IndexSet& s, int batch, Uninitialized) : context_(&context), batch_(batch)
  {
    hx_poly* p = nullptr;
    check(hx_poly_create_uninit(context.handle(), batch, s.data(), (int)s.size(), &p));
    h_.reset(p);
  }
  DoubleCRT(const DoubleCRT& other) : context_(other.context_), batch_(other.batch_)
  {
    // (a digit block lists its primes once per digit: hx_poly_copy takes over the source's row list,
    // the destination only has to exist)
    IndexSet s;
    for (int i : other.getIndexSet())
      if (std::find(s.begin(), s.end(), i) == s.end())
        s.push_back(i);
    hx_poly* p = nullptr;
    check(hx_poly_create_uninit(context_->handle(), batch_, s.data(), (int)s.size(), &p));
    h_.reset(p);
    check(hx_poly_copy(h_.get(), other.h_.get()));
  }
  DoubleCRT(DoubleCRT&&) = default;
  DoubleCRT& operator=(DoubleCRT&&) = default;
  DoubleCRT& operator=(const DoubleCRT& other)
  {
    if (this != &other) {
      if (context_ != other.context_)
        throw LogicError("DoubleCRT::operator=: incompatible objects");
      check(hx_poly_copy(h_.get(), other.h_.get()));
    }
    return *this;
  }

  const Context& getContext() const { return *context_; }
  IndexSet getIndexSet() const
  {
    int n = 0;
    check(hx_poly_shape(h_.get(), nullptr, &n, nullptr));
    IndexSet s((size_t)n);
    if (n)
      check(hx_poly_primes(h_.get(), s.data()));
    return s;
  }
  int batch() const { return batch_; }

  // rows in [row][batch][phi(m)] order
  void setRows(const std::vector<uint64_t>& rows) { check(hx_poly_upload(h_.get(), rows.data())); }
  std::vector<uint64_t> getRows() const
  {
    std::vector<uint64_t> out(getIndexSet().size() * (size_t)batch_ * (size_t)context_->getPhiM());
    check(hx_poly_download(h_.get(), out.data()));
    return out;
  }

  // DoubleCRT::randomize (src/DoubleCRT.cpp:1258-1378) on the device: uniform residues in every row
  // from the ChaCha20 stream (key, stream) -- see hx_randomize in helib_amd.h
  DoubleCRT& randomize(const uint8_t key32[32], uint64_t stream) { return chk(hx_randomize(h_.get(), key32, stream)); }

  // Cmodulus::FFT / iFFT on every row (coefficients <-> evaluations)
  DoubleCRT& FFT() { return chk(hx_ntt_forward(h_.get())); }
  DoubleCRT& iFFT() { return chk(hx_ntt_inverse(h_.get())); }

  // ring operations: require getIndexSet() <= other.getIndexSet(), else RuntimeError
  DoubleCRT& operator+=(const DoubleCRT& o) { return chk(hx_add(h_.get(), o.h_.get())); }
  DoubleCRT& operator-=(const DoubleCRT& o) { return chk(hx_sub(h_.get(), o.h_.get())); }
  DoubleCRT& operator*=(const DoubleCRT& o) { return chk(hx_mul(h_.get(), o.h_.get())); }
  DoubleCRT& Negate() { return chk(hx_negate(h_.get())); }
  // scalar given as its residues per row (the host reduces the ZZ modulo each prime)
  DoubleCRT& addConstant(const std::vector<uint64_t>& c) { return chk(hx_add_scalar(h_.get(), c.data())); }
  DoubleCRT& subConstant(const std::vector<uint64_t>& c) { return chk(hx_sub_scalar(h_.get(), c.data())); }
  DoubleCRT& mulConstant(const std::vector<uint64_t>& c) { return chk(hx_mul_scalar(h_.get(), c.data())); }
  DoubleCRT& operator*=(long num)
  {
    IndexSet s = getIndexSet();
    std::vector<uint64_t> c(s.size());
    for (size_t i = 0; i < s.size(); i++) {
      long q = context_->ithPrime(s[i]);
      long r = num % q;
      c[i] = (uint64_t)(r < 0 ? r + q : r);
    }
    return mulConstant(c);
  }
  // operator=(ZZ) with the per-row residues; Exp(e): entry-wise PowerMod
  DoubleCRT& setConstant(const std::vector<uint64_t>& c) { return chk(hx_set_scalar(h_.get(), c.data())); }
  DoubleCRT& Exp(long e)
  {
    if (e < 0)
      throw InvalidArgument("DoubleCRT::Exp: negative exponent");
    return chk(hx_exp(h_.get(), (uint64_t)e));
  }
  DoubleCRT& automorph(long k) { return chk(hx_automorph(h_.get(), (uint64_t)k)); }
  DoubleCRT& complexConj() { return chk(hx_complex_conj(h_.get())); }

  // prime-set operations
  DoubleCRT& removePrimes(const IndexSet& s) { return chk(hx_poly_remove_primes(h_.get(), s.data(), (int)s.size())); }
  DoubleCRT& addPrimes(const IndexSet& s) { return chk(hx_add_primes(h_.get(), s.data(), (int)s.size())); }
  // toPoly + PolyRed(t, abs=true) (the tail of SecKey::Decrypt): batch*phi(m) residues in [0,t)
  void toPolyMod(unsigned long t, unsigned long* out) const
  {
    check(hx_poly_rem(h_.get(), (uint64_t)t, reinterpret_cast<uint64_t*>(out)));
  }
  // DoubleCRT::toPoly(poly, s, positive) (src/DoubleCRT.cpp:925-1113): the integer polynomial of batch
  // element b modulo Q = the product of this object's primes that are in s (all of them by default):
  // inverse transform of a copy on the device, then the CRT reconstruction on the host -- coefficients
  // in [0, Q) when positive, else centred: v - Q for v >= (Q + 1) / 2 (:1053-1056, 1097-1098).
  // phi(m) coefficients, trailing zeros kept (the reference's ZZX is normalised; callers index by degree).
  std::vector<BigInt> toPoly(const IndexSet* s = nullptr, bool positive = false, int b = 0) const
  {
    if (b < 0 || b >= batch_)
      throw InvalidArgument("DoubleCRT::toPoly: batch element out of range");
    DoubleCRT tmp(*this);
    if (s) {
      IndexSet drop;
      for (int i : tmp.getIndexSet())
        if (std::find(s->begin(), s->end(), i) == s->end())
          drop.push_back(i);
      if (!drop.empty())
        tmp.removePrimes(drop);
    }
    const IndexSet idx = tmp.getIndexSet();
    const size_t L = idx.size(), n = (size_t)context_->getPhiM(), B = (size_t)batch_;
    std::vector<BigInt> out(n);
    if (L == 0)
      return out;
    tmp.iFFT();
    const std::vector<uint64_t> rows = tmp.getRows();
    auto mulmod = [](uint64_t a, uint64_t c, uint64_t q) { return (uint64_t)((unsigned __int128)a * c % q); };
    auto powmod = [&](uint64_t a, uint64_t e, uint64_t q) {
      uint64_t r = 1;
      for (a %= q; e; e >>= 1, a = mulmod(a, a, q))
        if (e & 1)
          r = mulmod(r, a, q);
      return r;
    };
    std::vector<uint64_t> q(L), inv(L * L, 0);
    for (size_t k = 0; k < L; k++)
      q[k] = (uint64_t)context_->ithPrime(idx[k]);
    for (size_t k = 0; k < L; k++)
      for (size_t l = 0; l < k; l++)
        inv[k * L + l] = powmod(q[l] % q[k], q[k] - 2, q[k]);   // q_l^-1 mod q_k
    BigInt Q(1);
    for (size_t k = 0; k < L; k++)
      Q.mulAdd(q[k], 0);
    const BigInt half = Q.halfUp();
    std::vector<uint64_t> a(L);
    for (size_t j = 0; j < n; j++) {
      // Garner: v = a_0 + a_1 q_0 + a_2 q_0 q_1 + ...  with 0 <= a_k < q_k
      for (size_t k = 0; k < L; k++) {
        uint64_t x = rows[(k * B + (size_t)b) * n + j];
        for (size_t l = 0; l < k; l++) {
          const uint64_t al = a[l] % q[k];
          x = mulmod(x >= al ? x - al : x + q[k] - al, inv[k * L + l], q[k]);
        }
        a[k] = x;
      }
      BigInt v(a[L - 1]);
      for (size_t k = L - 1; k-- > 0;)
        v.mulAdd(q[k], a[k]);
      if (!positive && BigInt::cmpMag(v, half) >= 0)
        v = BigInt::subMag(Q, v).negated();
      out[j] = std::move(v);
    }
    return out;
  }
  DoubleCRT& addPrimesAndScale(const IndexSet& s)
  {
    return chk(hx_add_primes_and_scale(h_.get(), s.data(), (int)s.size()));
  }
  // scaleDownToSet(s, ptxtSpace): keep only the primes in s
  DoubleCRT& scaleDownToSet(const IndexSet& s, long ptxtSpace)
  {
    IndexSet drop;
    for (int i : getIndexSet()) {
      bool keep = false;
      for (int j : s)
        keep |= (i == j);
      if (!keep)
        drop.push_back(i);
    }
    return chk(hx_scale_down(h_.get(), drop.data(), (int)drop.size(), (uint64_t)ptxtSpace));
  }
  // breakIntoDigits: digit d = digits[d] (prime indices), special primes appended to every digit;
  // result: one object with digits.size()*(rows+|special|) rows, block d = digit d
  // norms (optional): digits.size()*batch values embeddingLargestCoeff(digit d of element b) / P_d, the
  // pieces of the reference's return value (src/DoubleCRT.cpp:538-545), measured on the device
  // (an array of digits.size()*batch doubles; filled on return, or -- Context::deferNorms(true) -- at the next
  // flush: take it from Context::normBuffer then)
  DoubleCRT breakIntoDigits(const std::vector<IndexSet>& digits, const IndexSet& special, double* norms = nullptr) const
  {
    std::vector<int> idx, off(1, 0);
    for (auto& d : digits) {
      idx.insert(idx.end(), d.begin(), d.end());
      off.push_back((int)idx.size());
    }
    DoubleCRT out(*context_, getIndexSet(), batch_);
    if (norms) {
      check(hx_break_into_digits_norms(h_.get(), idx.data(), off.data(), (int)digits.size(), special.data(),
                                       (int)special.size(), out.h_.get(), norms));
    } else {
      check(hx_break_into_digits(h_.get(), idx.data(), off.data(), (int)digits.size(), special.data(),
                                 (int)special.size(), out.h_.get()));
    }
    return out;
  }

  hx_poly* handle() const { return h_.get(); }

private:
  struct Del {
    void operator()(hx_poly* p) const { hx_poly_destroy(p); }
  };
  DoubleCRT& chk(int rc)
  {
    check(rc);
    return *this;
  }
  const Context* context_;
  int batch_;
  std::unique_ptr<hx_poly, Del> h_;
};

class KeySwitch {
public:
  // b, a: [ndig][rows][phi(m)] on primes `rows` (ctxt primes followed by special primes)
  KeySwitch(const Context& c, int ndig, const IndexSet& rows, const std::vector<uint64_t>& b,
            const std::vector<uint64_t>& a)
  {
    hx_ksk* k = nullptr;
    check(hx_ksk_create(c.handle(), ndig, rows.data(), (int)rows.size(), b.data(), a.data(), &k));
    h_.reset(k);
    rows_ = rows;
  }
  hx_ksk* handle() const { return h_.get(); }
  const IndexSet& rows() const { return rows_; }
  int ndig() const
  {
    int d = 0, n = 0;
    check(hx_ksk_shape(h_.get(), &d, &n, nullptr));
    return d;
  }
  // the matrix back on the host, [ndig][rows][phi(m)] each (what the constructor took)
  void download(std::vector<uint64_t>& b, std::vector<uint64_t>& a, size_t phim) const
  {
    const size_t words = (size_t)ndig() * rows_.size() * phim;
    b.resize(words);
    a.resize(words);
    check(hx_ksk_download(h_.get(), b.data(), a.data()));
  }
  // the matrix' rows are exactly `first` followed by `then`, in this order (what hx_mul_relin expects of the
  // ciphertext's primes and the special primes)
  bool coversInOrder(const IndexSet& first, const IndexSet& then) const
  {
    if (rows_.size() != first.size() + then.size())
      return false;
    for (size_t i = 0; i < first.size(); i++)
      if (rows_[i] != first[i])
        return false;
    for (size_t i = 0; i < then.size(); i++)
      if (rows_[first.size() + i] != then[i])
        return false;
    return true;
  }

private:
  struct Del {
    void operator()(hx_ksk* p) const { hx_ksk_destroy(p); }
  };
  std::unique_ptr<hx_ksk, Del> h_;
  IndexSet rows_;
};

// Cmodulus (include/helib/CModulus.h:56-145): one modulus q with the tables for FFT / iFFT modulo q over
// Z_m^*.  The reference's constructor takes (PAlgebra, q, root); this one takes m and builds its own
// one-prime device context.  root: the 2m-th (m even, incl. powers of two: NTL's RootTable[0][k]) or m-th
// (m odd) root of unity the rows are defined by; 0 = FindPrimRootT(q, e) (src/CModulus.cpp:148-164).
// FFT / iFFT move one row over PCIe per call -- the reference's own granularity, kept for code written
// against Cmodulus; whole DoubleCRT objects stay on the device (class DoubleCRT above).
class Cmodulus {
public:
  Cmodulus(unsigned long m, long q, long root = 0, int device = 0) : ctx_(std::make_shared<Context>((uint64_t)m, device))
  {
    if (q <= 1)
      throw InvalidArgument("Cmodulus: q must be a prime");
    ctx_->addPrime((uint64_t)q, (uint64_t)(root < 0 ? 0 : root));
    row_ = std::make_shared<DoubleCRT>(*ctx_, IndexSet{0}, 1);
  }
  unsigned long getM() const { return (unsigned long)ctx_->getM(); }
  unsigned long getPhiM() const { return (unsigned long)ctx_->getPhiM(); }
  long getQ() const { return ctx_->ithPrime(0); }
  long getRoot() const { return (long)ctx_->ithRoot(0); }
  const Context& getContext() const { return *ctx_; }
  // y = FFT(x): x = coefficients (degree < phi(m); any sign, reduced modulo q here as the reference's
  // conv(ZZX -> zz_pX) does, src/CModulus.cpp:446-484); y[j] = x(zeta^{t_j}), t_j the j-th element of Z_m^*
  void FFT(std::vector<long>& y, const std::vector<long>& x) const
  {
    const size_t n = getPhiM();
    if (x.size() > n)
      throw InvalidArgument("Cmodulus::FFT: polynomial of degree >= phi(m)");
    const long q = getQ();
    std::vector<uint64_t> rows(n, 0);
    for (size_t i = 0; i < x.size(); i++) {
      long r = x[i] % q;
      rows[i] = (uint64_t)(r < 0 ? r + q : r);
    }
    std::lock_guard<std::mutex> lock(*mu_);
    row_->setRows(rows);
    row_->FFT();
    rows = row_->getRows();
    y.assign(rows.begin(), rows.end());
  }
  // x = FFT^{-1}(y): phi(m) coefficients in [0, q) (src/CModulus.cpp:486-578, incl. rem Phi_m for general m)
  void iFFT(std::vector<long>& x, const std::vector<long>& y) const
  {
    const size_t n = getPhiM();
    if (y.size() != n)
      throw InvalidArgument("Cmodulus::iFFT: y must have phi(m) entries");
    const long q = getQ();
    std::vector<uint64_t> rows(n);
    for (size_t i = 0; i < n; i++) {
      if (y[i] < 0 || y[i] >= q)
        throw InvalidArgument("Cmodulus::iFFT: entries must be in [0, q)");
      rows[i] = (uint64_t)y[i];
    }
    std::lock_guard<std::mutex> lock(*mu_);
    row_->setRows(rows);
    row_->iFFT();
    rows = row_->getRows();
    x.assign(rows.begin(), rows.end());
  }

private:
  std::shared_ptr<Context> ctx_;
  std::shared_ptr<DoubleCRT> row_;                         // the one-row work object (copies of a Cmodulus share it)
  std::shared_ptr<std::mutex> mu_ = std::make_shared<std::mutex>();
};

inline void flatten(const std::vector<IndexSet>& digits, std::vector<int>& idx, std::vector<int>& off)
{
  idx.clear();
  off.assign(1, 0);
  for (auto& d : digits) {
    idx.insert(idx.end(), d.begin(), d.end());
    off.push_back((int)idx.size());
  }
}

// Ctxt::tensorProduct for two 2-part ciphertexts
inline void tensorProduct(const DoubleCRT& c0, const DoubleCRT& c1, const DoubleCRT& d0, const DoubleCRT& d1,
                          DoubleCRT& o0, DoubleCRT& o1, DoubleCRT& o2)
{
  check(hx_tensor(c0.handle(), c1.handle(), d0.handle(), d1.handle(), o0.handle(), o1.handle(), o2.handle()));
}
// Ctxt::keySwitchDigits
inline void keySwitchDigits(const KeySwitch& W, const DoubleCRT& digits, DoubleCRT& partOne, DoubleCRT& partS)
{
  check(hx_key_switch_digits(digits.handle(), W.handle(), partOne.handle(), partS.handle()));
}
// Ctxt::multiplyBy data path at a fixed level (tensorProduct + reLinearize)
inline void multiplyBy(const DoubleCRT& c0, const DoubleCRT& c1, const DoubleCRT& d0, const DoubleCRT& d1,
                       const KeySwitch& W, const std::vector<IndexSet>& digits, DoubleCRT& out0, DoubleCRT& out1)
{
  std::vector<int> idx, off;
  flatten(digits, idx, off);
  check(hx_mul_relin(c0.handle(), c1.handle(), d0.handle(), d1.handle(), W.handle(), idx.data(), off.data(),
                     (int)digits.size(), out0.handle(), out1.handle()));
}
// Ctxt::reLinearize for parts (1, s, s^2)
inline void reLinearize(const DoubleCRT& t0, const DoubleCRT& t1, const DoubleCRT& t2, const KeySwitch& W,
                        const std::vector<IndexSet>& digits, const IndexSet& special, DoubleCRT& out0,
                        DoubleCRT& out1)
{
  std::vector<int> idx, off;
  flatten(digits, idx, off);
  check(hx_relinearize(t0.handle(), t1.handle(), t2.handle(), W.handle(), idx.data(), off.data(),
                       (int)digits.size(), special.data(), (int)special.size(), out0.handle(), out1.handle()));
}

}  // namespace helib_amd

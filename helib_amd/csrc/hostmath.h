// hostmath.h -- exact host-side number theory used when the library builds its
// device tables (one-time, per prime / per prime set).  Product code: it must
// not depend on oracle/.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace hxh {

typedef unsigned __int128 u128;

inline uint64_t mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
inline uint64_t addmod(uint64_t a, uint64_t b, uint64_t q)
{
  uint64_t s = a + b;
  return s >= q ? s - q : s;
}
inline uint64_t submod(uint64_t a, uint64_t b, uint64_t q) { return a >= b ? a - b : a + q - b; }
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
// modular inverse for any modulus (0 when not invertible)
inline uint64_t invmod(uint64_t a, uint64_t q)
{
  __int128 t = 0, nt = 1, r = (__int128)q, nr = (__int128)(a % q);
  while (nr != 0) {
    __int128 quo = r / nr, tmp = t - quo * nt;
    t = nt;
    nt = tmp;
    tmp = r - quo * nr;
    r = nr;
    nr = tmp;
  }
  if (r != 1)
    return 0;
  if (t < 0)
    t += q;
  return (uint64_t)t;
}
inline uint64_t shoup(uint64_t w, uint64_t q) { return (uint64_t)((((u128)w) << 64) / q); }
inline uint64_t gcd(uint64_t a, uint64_t b)
{
  while (b) {
    uint64_t t = a % b;
    a = b;
    b = t;
  }
  return a;
}
inline int bitlen(uint64_t x)
{
  int n = 0;
  while (x) {
    n++;
    x >>= 1;
  }
  return n;
}

// Deterministic Miller-Rabin, n < 2^64.
inline bool is_prime(uint64_t n)
{
  static const uint64_t B[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2)
    return false;
  for (uint64_t b : B) {
    if (n == b)
      return true;
    if (n % b == 0)
      return false;
  }
  uint64_t d = n - 1;
  int s = 0;
  while (!(d & 1)) {
    d >>= 1;
    s++;
  }
  for (uint64_t b : B) {
    uint64_t x = powmod(b, d, n);
    if (x == 1 || x == n - 1)
      continue;
    bool comp = true;
    for (int r = 1; r < s; r++) {
      x = mulmod(x, x, n);
      if (x == n - 1) {
        comp = false;
        break;
      }
    }
    if (comp)
      return false;
  }
  return true;
}

// Deterministic e-th root of unity mod prime q: the rule of HElib's
// FindPrimRootT (src/NumbTh.cpp:436-493): for each prime p | e take the
// smallest prime s with s^((q-1)/p) != 1 and contribute s^((q-1)/p^v).
inline uint64_t find_prim_root(uint64_t q, uint64_t e)
{
  if (e == 0 || (q - 1) % e != 0)
    return 0;
  std::vector<uint64_t> facts;
  uint64_t x = e;
  for (uint64_t p = 2; p * p <= x; p++)
    if (x % p == 0) {
      facts.push_back(p);
      while (x % p == 0)
        x /= p;
    }
  if (x > 1)
    facts.push_back(x);
  uint64_t root = 1;
  for (uint64_t p : facts) {
    uint64_t pp = p, ee = e / p;
    while (ee % p == 0) {
      ee /= p;
      pp *= p;
    }
    uint64_t s = 1, t;
    do {
      do {
        s++;
      } while (!is_prime(s));
      t = powmod(s, (q - 1) / p, q);
    } while (t == 1);
    root = mulmod(root, powmod(s, (q - 1) / pp, q), q);
  }
  return root;
}

// The root Intel HEXL's NTT(degree, q) object picks for itself (HEXL 1.2.1, hexl/number-theory:
// MinimalPrimitiveRoot(2*degree, q)): the SMALLEST primitive e-th root of unity mod q, e a power of
// two.  HEXL starts from a random primitive root and walks its odd powers keeping the minimum; the
// minimum does not depend on the start, so any primitive e-th root serves as one.
inline uint64_t hexl_minimal_primitive_root(uint64_t q, uint64_t e)
{
  uint64_t g = find_prim_root(q, e);
  if (g == 0)
    return 0;
  const uint64_t g2 = mulmod(g, g, q);
  uint64_t cur = g, best = g;
  for (uint64_t i = 0; i < e / 2; i++) {  // the e/2 odd powers = every primitive e-th root
    if (cur < best)
      best = cur;
    cur = mulmod(cur, g2, q);
  }
  return best;
}

// ---- minimal unsigned big integer (little-endian 64-bit limbs) ----
struct BigU {
  std::vector<uint64_t> d;
  explicit BigU(uint64_t v = 0) : d(1, v) {}
  void mul_word(uint64_t w)
  {
    uint64_t c = 0;
    for (auto& x : d) {
      u128 p = (u128)x * w + c;
      x = (uint64_t)p;
      c = (uint64_t)(p >> 64);
    }
    if (c)
      d.push_back(c);
  }
  uint64_t divmod_word(uint64_t w)  // *this /= w, returns remainder
  {
    u128 r = 0;
    for (size_t i = d.size(); i-- > 0;) {
      u128 cur = (r << 64) | d[i];
      d[i] = (uint64_t)(cur / w);
      r = cur % w;
    }
    while (d.size() > 1 && d.back() == 0)
      d.pop_back();
    return (uint64_t)r;
  }
  uint64_t mod_word(uint64_t w) const
  {
    u128 r = 0;
    for (size_t i = d.size(); i-- > 0;)
      r = ((r << 64) | d[i]) % w;
    return (uint64_t)r;
  }
  void sub_word(uint64_t w)  // assumes *this >= w
  {
    size_t i = 0;
    while (w) {
      uint64_t old = d[i];
      d[i] = old - w;
      w = old < w ? 1 : 0;
      i++;
    }
  }
  void shr1()
  {
    for (size_t i = 0; i < d.size(); i++) {
      uint64_t hi = i + 1 < d.size() ? d[i + 1] : 0;
      d[i] = (d[i] >> 1) | (hi << 63);
    }
    while (d.size() > 1 && d.back() == 0)
      d.pop_back();
  }
  bool is_zero() const { return d.size() == 1 && d[0] == 0; }
};

}  // namespace hxh

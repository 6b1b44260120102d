// helib_amd_wire.hpp -- header-only C++17: HElib 2.2.0's binary wire formats for the objects of this
// path, host side only -- no device call in this file.
//
//   Rows         DoubleCRT::writeTo / read        src/DoubleCRT.cpp:1530-1561 (IndexSet :288-297 of
//                                                 src/IndexSet.cpp, write_ntl_vec_long src/binio.cpp:103-146)
//   CtxtDesc     Ctxt::writeTo / read             src/Ctxt.cpp:2584-2640 (SerializeHeader src/binio.h:88-146,
//                                                 xdouble = double mantissa + int64 exponent, src/binio.cpp:165-178)
//   KeySwitchDesc KeySwitch::writeTo / readFrom   src/keySwitching.cpp:196-241 (prgSeed: write_raw_ZZ)
//   ContextDesc  Context::writeTo / readParamsFrom   src/Context.cpp:324-442
//   PubKeyDesc / SecKeyDesc  PubKey::writeTo :888-974, SecKey::writeTo :1736-1784 of src/keys.cpp
//
// legacy = true is the layout of the reference's own fixture tests/test_resources/iotest_bin*.bin
// (written by an older HElib): no SerializeHeader, no intFactor / ptxtMag / ratFactor in a Ctxt, no
// noiseBound in a KeySwitch, the context as a "|BS[" base block + a shorter "|CN[" block, the public
// key holding the base block only and integer Hamming weights where 2.2.0 has double bounds.
// Errors are IOError (std::runtime_error), with the reference's messages where it has one.
//
// The descriptions are plain data; fromPoly / toRows move rows between them and anything with
// getIndexSet() / getRows() / setRows() (helib_amd::DoubleCRT), so a Ctxt or a key of this engine
// goes to and from the bytes a HElib build reads and writes.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace helib_amd {
namespace wire {

struct IOError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct XDouble {  // NTL::xdouble: mantissa * 2^(114 * exponent)
  double mantissa = 0;
  int64_t exponent = 0;
  bool operator==(const XDouble& o) const { return mantissa == o.mantissa && exponent == o.exponent; }
};

struct Rows {  // one DoubleCRT: prime indices ascending, data[r * n + j]
  std::vector<long> idx;
  size_t n = 0;
  std::vector<uint64_t> data;
};
struct Part {
  Rows rows;
  long handle[3] = {0, 1, 0};  // powerOfS, powerOfX, secretKeyID
};
struct CtxtDesc {
  long ptxtSpace = 0, intFactor = 1;
  XDouble ptxtMag{1.0, 0}, ratFactor{1.0, 0}, noiseBound{0.0, 0};
  std::vector<long> primeSet;
  std::vector<Part> parts;
};
struct KeySwitchDesc {
  long fromKey[3] = {0, 1, 0};
  long toKeyID = 0, ptxtSpace = 0;
  std::vector<Rows> b;
  std::vector<uint8_t> prgSeed;  // little-endian magnitude bytes (write_raw_ZZ)
  XDouble noiseBound{0.0, 0};
};
struct ContextDesc {
  long p = 0, r = 0, m = 0;
  std::vector<long> gens, ords;
  XDouble stdev{3.2, 0};
  double scale = 10.0;
  std::vector<long> smallPrimes, specialPrimes, qs;
  std::vector<std::vector<long>> digits;
  long hwt_param = 0, e_param = 0, ePrime_param = 0;
  std::vector<long> mvec;
  long build_cache = 0, alsoThick = 0;
};
struct PubKeyDesc {
  ContextDesc context;
  CtxtDesc pubEncrKey;
  std::vector<double> skBounds;   // legacy: the integer Hamming weights, as doubles
  std::vector<KeySwitchDesc> keySwitching;
  std::vector<std::vector<long>> keySwitchMap;
  std::vector<long> KS_strategy;
  long recryptKeyID = -1;
  CtxtDesc recryptEkey;
};
struct SecKeyDesc : PubKeyDesc {
  std::vector<Rows> sKeys;
};

enum StructId { ID_CONTEXT = 5, ID_PUBKEY = 10, ID_SECKEY = 15, ID_CTXT = 20 };

// ---------------------------------------------------------------- byte sink / source
class Writer {
public:
  std::string out;
  void raw(const void* p, size_t n) { out.append(static_cast<const char*>(p), n); }
  void i64(int64_t v) { raw(&v, 8); }       // write_raw_int: little-endian 8 bytes
  void i32(int32_t v) { raw(&v, 4); }
  void f64(double v) { raw(&v, 8); }        // write_raw_double: the bits of the double
  void xd(const XDouble& x)
  {
    f64(x.mantissa);
    i64(x.exponent);
  }
  void eye(const char* tag) { raw(tag, 4); }
  void header(int structId)  // SerializeHeader<T>: 24 bytes
  {
    const unsigned char h[24] = {'|', 'H', 'E', '[', 0, 0, 1, 0, 2, 2, 0, 0, (unsigned char)structId,
                                 0,   0,   0,   0,   0, 0, 0, ']', 'H', 'E', '|'};
    raw(h, 24);
  }
  void longs(const std::vector<long>& v)  // write_raw_vector<long> / IndexSet::writeTo
  {
    i64((int64_t)v.size());
    for (long x : v)
      i64(x);
  }
  void vecLong(const std::vector<long>& v)  // write_ntl_vec_long, 64-bit words
  {
    i32((int32_t)v.size());
    i32(8);
    for (long x : v)
      i64(x);
  }
};

class Reader {
public:
  Reader(const void* data, size_t size) : p_(static_cast<const unsigned char*>(data)), n_(size) {}
  size_t pos = 0;
  bool done() const { return pos == n_; }
  void raw(void* dst, size_t n)
  {
    if (n > n_ - pos)
      throw IOError("unexpected end of stream");
    std::memcpy(dst, p_ + pos, n);
    pos += n;
  }
  int64_t i64()
  {
    int64_t v;
    raw(&v, 8);
    return v;
  }
  int32_t i32()
  {
    int32_t v;
    raw(&v, 4);
    return v;
  }
  double f64()
  {
    double v;
    raw(&v, 8);
    return v;
  }
  XDouble xd()
  {
    XDouble x;
    x.mantissa = f64();
    x.exponent = i64();
    return x;
  }
  void eye(const char* tag, const char* what)
  {
    char t[4];
    raw(t, 4);
    if (std::memcmp(t, tag, 4) != 0)
      throw IOError(std::string("Could not find ") + what + " eye catcher");
  }
  void header(int structId)
  {
    unsigned char h[24];
    raw(h, 24);
    if (std::memcmp(h, "|HE[", 4) != 0 || std::memcmp(h + 20, "]HE|", 4) != 0)
      throw IOError("Eye catchers for header mismatch");
    const unsigned char v[4] = {0, 0, 1, 0};
    if (std::memcmp(h + 4, v, 4) != 0)
      throw IOError("Header: version not supported");
    if (h[12] != (unsigned char)structId)
      throw IOError("Header: wrong structId");
  }
  std::vector<long> longs()
  {
    int64_t n = i64();
    if (n < 0 || (uint64_t)n > (n_ - pos) / 8)
      throw IOError("implausible vector length");
    std::vector<long> v((size_t)n);
    for (auto& x : v)
      x = (long)i64();
    return v;
  }
  std::vector<long> vecLong()
  {
    int32_t n = i32(), sz = i32();
    if (n < 0 || (sz != 4 && sz != 8))
      throw IOError("intSize must be 32 or 64 bit for binary IO");
    std::vector<long> v((size_t)n);
    for (auto& x : v)
      x = sz == 8 ? (long)i64() : (long)i32();
    return v;
  }

private:
  const unsigned char* p_;
  size_t n_;
};

// ---------------------------------------------------------------- DoubleCRT
inline void write(Writer& w, const Rows& r)
{
  if (r.data.size() != r.idx.size() * r.n)
    throw IOError("one row of n words per prime index");
  w.longs(r.idx);
  for (size_t k = 0; k < r.idx.size(); k++) {
    w.i32((int32_t)r.n);
    w.i32(8);
    w.raw(r.data.data() + k * r.n, r.n * 8);
  }
}
inline Rows readRows(Reader& rd)
{
  Rows r;
  r.idx = rd.longs();
  // IndexSet::readFrom inserts what it reads into a set: the rows follow in ascending order
  for (size_t i = 1; i < r.idx.size(); i++)
    for (size_t j = i; j > 0 && r.idx[j] < r.idx[j - 1]; j--)
      std::swap(r.idx[j], r.idx[j - 1]);
  size_t u = 0;
  for (size_t i = 0; i < r.idx.size(); i++)
    if (i == 0 || r.idx[i] != r.idx[u - 1])
      r.idx[u++] = r.idx[i];
  r.idx.resize(u);
  for (size_t k = 0; k < r.idx.size(); k++) {
    int32_t n = rd.i32(), sz = rd.i32();
    if (n < 0 || (sz != 4 && sz != 8))
      throw IOError("intSize must be 32 or 64 bit for binary IO");
    if (k == 0)
      r.n = (size_t)n;
    else if ((size_t)n != r.n)
      throw IOError("rows of unequal length");
    for (int32_t j = 0; j < n; j++)
      r.data.push_back(sz == 8 ? (uint64_t)rd.i64() : (uint64_t)(uint32_t)rd.i32());
  }
  return r;
}

// ---------------------------------------------------------------- Ctxt
inline void write(Writer& w, const CtxtDesc& c, bool legacy = false)
{
  if (!legacy)
    w.header(ID_CTXT);
  w.eye("|CX[");
  w.i64(c.ptxtSpace);
  if (!legacy) {
    w.i64(c.intFactor);
    w.xd(c.ptxtMag);
    w.xd(c.ratFactor);
  }
  w.xd(c.noiseBound);
  w.longs(c.primeSet);
  w.i64((int64_t)c.parts.size());
  for (auto& p : c.parts) {
    write(w, p.rows);
    for (long h : p.handle)
      w.i64(h);
  }
  w.eye("]CX|");
}
inline CtxtDesc readCtxt(Reader& rd, bool legacy = false)
{
  CtxtDesc c;
  if (!legacy)
    rd.header(ID_CTXT);
  rd.eye("|CX[", "pre-ciphertext");
  c.ptxtSpace = (long)rd.i64();
  if (!legacy) {
    c.intFactor = (long)rd.i64();
    c.ptxtMag = rd.xd();
    c.ratFactor = rd.xd();
  }
  c.noiseBound = rd.xd();
  c.primeSet = rd.longs();
  int64_t np = rd.i64();
  if (np < 0 || np > 1024)
    throw IOError("implausible number of ciphertext parts");
  for (int64_t i = 0; i < np; i++) {
    Part p;
    p.rows = readRows(rd);
    for (long& h : p.handle)
      h = (long)rd.i64();
    c.parts.push_back(std::move(p));
  }
  rd.eye("]CX|", "post-ciphertext");
  return c;
}

// ---------------------------------------------------------------- KeySwitch
inline void write(Writer& w, const KeySwitchDesc& k, bool legacy = false)
{
  if (k.prgSeed.empty())
    throw IOError("Number of bytes to write must be non-negative");  // write_raw_ZZ of 0
  w.eye("|KM[");
  for (long h : k.fromKey)
    w.i64(h);
  w.i64(k.toKeyID);
  w.i64(k.ptxtSpace);
  w.i64((int64_t)k.b.size());
  for (auto& r : k.b)
    write(w, r);
  w.i64((int64_t)k.prgSeed.size());
  w.raw(k.prgSeed.data(), k.prgSeed.size());
  if (!legacy)
    w.xd(k.noiseBound);
  w.eye("]KM|");
}
inline KeySwitchDesc readKeySwitch(Reader& rd, bool legacy = false)
{
  KeySwitchDesc k;
  rd.eye("|KM[", "pre-key-switching-matrix");
  for (long& h : k.fromKey)
    h = (long)rd.i64();
  k.toKeyID = (long)rd.i64();
  k.ptxtSpace = (long)rd.i64();
  int64_t nb = rd.i64();
  if (nb < 0 || nb > 4096)
    throw IOError("implausible number of key-switching columns");
  for (int64_t i = 0; i < nb; i++)
    k.b.push_back(readRows(rd));
  int64_t bytes = rd.i64();
  if (bytes <= 0 || bytes > 1 << 20)
    throw IOError("Number of bytes to write must be non-negative");
  k.prgSeed.resize((size_t)bytes);
  rd.raw(k.prgSeed.data(), (size_t)bytes);
  if (!legacy)
    k.noiseBound = rd.xd();
  rd.eye("]KM|", "post-key-switching-matrix");
  return k;
}

// ---------------------------------------------------------------- Context
inline void writeBase(Writer& w, const ContextDesc& c)
{
  w.i64(c.p);
  w.i64(c.r);
  w.i64(c.m);
  w.longs(c.gens);
  w.longs(c.ords);
}
inline void readBase(Reader& rd, ContextDesc& c)
{
  c.p = (long)rd.i64();
  c.r = (long)rd.i64();
  c.m = (long)rd.i64();
  c.gens = rd.longs();
  c.ords = rd.longs();
}
inline void writeContextBase(Writer& w, const ContextDesc& c)  // the legacy "|BS[" block
{
  w.eye("|BS[");
  writeBase(w, c);
  w.eye("]BS|");
}
inline void write(Writer& w, const ContextDesc& c, bool legacy = false)
{
  if (legacy) {
    writeContextBase(w, c);
    w.eye("|CN[");
    w.f64(c.stdev.mantissa);  // a plain double in the old layout
  } else {
    w.header(ID_CONTEXT);
    w.eye("|CN[");
    writeBase(w, c);
    w.xd(c.stdev);
    w.f64(c.scale);
  }
  w.longs(c.smallPrimes);
  w.longs(c.specialPrimes);
  w.longs(c.qs);
  w.i64((int64_t)c.digits.size());
  for (auto& d : c.digits)
    w.longs(d);
  if (!legacy) {
    w.i64(c.hwt_param);
    w.i64(c.e_param);
    w.i64(c.ePrime_param);
  }
  w.vecLong(c.mvec);
  w.i64(c.build_cache);
  w.i64(c.alsoThick);
  w.eye("]CN|");
}
inline ContextDesc readContextBase(Reader& rd)
{
  ContextDesc c;
  rd.eye("|BS[", "pre-context-base");
  readBase(rd, c);
  rd.eye("]BS|", "post-context-base");
  return c;
}
inline ContextDesc readContext(Reader& rd, bool legacy = false)
{
  ContextDesc c;
  if (legacy) {
    c = readContextBase(rd);
    rd.eye("|CN[", "pre-context");
    c.stdev = XDouble{rd.f64(), 0};
  } else {
    rd.header(ID_CONTEXT);
    rd.eye("|CN[", "pre-context");
    readBase(rd, c);
    c.stdev = rd.xd();
    c.scale = rd.f64();
  }
  c.smallPrimes = rd.longs();
  c.specialPrimes = rd.longs();
  c.qs = rd.longs();
  int64_t nd = rd.i64();
  if (nd < 0 || nd > 4096)
    throw IOError("implausible number of digits");
  for (int64_t i = 0; i < nd; i++)
    c.digits.push_back(rd.longs());
  if (!legacy) {
    c.hwt_param = (long)rd.i64();
    c.e_param = (long)rd.i64();
    c.ePrime_param = (long)rd.i64();
  }
  c.mvec = rd.vecLong();
  c.build_cache = (long)rd.i64();
  c.alsoThick = (long)rd.i64();
  rd.eye("]CN|", "post-context");
  return c;
}
inline bool sameContext(const ContextDesc& a, const ContextDesc& b, bool baseOnly = false)
{
  bool base = a.p == b.p && a.r == b.r && a.m == b.m && a.gens == b.gens && a.ords == b.ords;
  if (baseOnly)
    return base;
  return base && a.qs == b.qs && a.smallPrimes == b.smallPrimes && a.specialPrimes == b.specialPrimes &&
         a.digits == b.digits && a.stdev == b.stdev;
}

// ---------------------------------------------------------------- PubKey / SecKey
inline void write(Writer& w, const PubKeyDesc& k, bool legacy = false)
{
  if (!legacy)
    w.header(ID_PUBKEY);
  w.eye("|PK[");
  if (legacy)
    writeContextBase(w, k.context);
  else
    write(w, k.context);
  write(w, k.pubEncrKey, legacy);
  w.i64((int64_t)k.skBounds.size());
  for (double b : k.skBounds) {
    if (legacy)
      w.i64((int64_t)b);
    else
      w.f64(b);
  }
  w.i64((int64_t)k.keySwitching.size());
  for (auto& m : k.keySwitching)
    write(w, m, legacy);
  w.i64((int64_t)k.keySwitchMap.size());
  for (auto& v : k.keySwitchMap)
    w.longs(v);
  w.vecLong(k.KS_strategy);
  w.i64(k.recryptKeyID);
  write(w, k.recryptEkey, legacy);
  w.eye("]PK|");
}
inline void readPubKeyInto(Reader& rd, PubKeyDesc& k, bool legacy, const ContextDesc* expect)
{
  if (!legacy)
    rd.header(ID_PUBKEY);
  rd.eye("|PK[", "pre-public key");
  k.context = legacy ? readContextBase(rd) : readContext(rd);
  if (expect && !sameContext(k.context, *expect, legacy))
    throw IOError("Context mismatch");
  k.pubEncrKey = readCtxt(rd, legacy);
  int64_t n = rd.i64();
  if (n < 0 || n > 4096)
    throw IOError("implausible number of secret keys");
  for (int64_t i = 0; i < n; i++)
    k.skBounds.push_back(legacy ? (double)rd.i64() : rd.f64());
  n = rd.i64();
  if (n < 0 || n > 1 << 20)
    throw IOError("implausible number of key-switching matrices");
  for (int64_t i = 0; i < n; i++)
    k.keySwitching.push_back(readKeySwitch(rd, legacy));
  n = rd.i64();
  if (n < 0 || n > 4096)
    throw IOError("implausible key-switch map");
  for (int64_t i = 0; i < n; i++)
    k.keySwitchMap.push_back(rd.longs());
  k.KS_strategy = rd.vecLong();
  k.recryptKeyID = (long)rd.i64();
  k.recryptEkey = readCtxt(rd, legacy);
  rd.eye("]PK|", "post-public key");
}
inline PubKeyDesc readPubKey(Reader& rd, bool legacy = false, const ContextDesc* expect = nullptr)
{
  PubKeyDesc k;
  readPubKeyInto(rd, k, legacy, expect);
  return k;
}
inline void write(Writer& w, const SecKeyDesc& k, bool legacy = false, bool sk_only = false)
{
  if (!legacy)
    w.header(ID_SECKEY);
  w.eye("|SK[");
  if (sk_only)
    write(w, k.context, legacy);
  else
    write(w, static_cast<const PubKeyDesc&>(k), legacy);
  w.i64((int64_t)k.sKeys.size());
  for (auto& r : k.sKeys)
    write(w, r);
  w.eye("]SK|");
}
inline SecKeyDesc readSecKey(Reader& rd, bool legacy = false, bool sk_only = false,
                             const ContextDesc* expect = nullptr)
{
  SecKeyDesc k;
  if (!legacy)
    rd.header(ID_SECKEY);
  rd.eye("|SK[", "pre-secret key");
  if (sk_only) {
    k.context = readContext(rd, legacy);
    if (expect && !sameContext(k.context, *expect))
      throw IOError("Context mismatch");
  } else {
    readPubKeyInto(rd, k, legacy, expect);
  }
  int64_t n = rd.i64();
  if (n < 0 || n > 4096)
    throw IOError("implausible number of secret keys");
  for (int64_t i = 0; i < n; i++)
    k.sKeys.push_back(readRows(rd));
  rd.eye("]SK|", "post-secret key");
  return k;
}

// PubKey::setKeySwitchMap in the stored form (src/keys.cpp:122-172): entry k = the INDEX in
// keySwitching of the matrix for the first step of X -> X^k, -1 if unreachable
inline std::vector<long> keySwitchMapOf(long m, const std::vector<KeySwitchDesc>& ks, long keyId = 0)
{
  std::vector<std::pair<long, long>> edges;
  for (size_t i = 0; i < ks.size(); i++)
    if (ks[i].toKeyID == keyId && ks[i].fromKey[0] == 1 && ks[i].fromKey[2] == keyId)
      edges.emplace_back(ks[i].fromKey[1], (long)i);
  std::vector<long> map((size_t)m, -1), queue{1};
  for (size_t head = 0; head < queue.size(); head++) {
    long cur = queue[head];
    for (auto& e : edges) {
      long nxt = (long)((unsigned __int128)cur * (unsigned long)e.first % (unsigned long)m);
      if (map[(size_t)nxt] == -1) {
        map[(size_t)nxt] = e.second;
        queue.push_back(nxt);
      }
    }
  }
  return map;
}

// ---------------------------------------------------------------- to / from polynomial objects
// Poly: anything with getIndexSet() (a sequence of prime indices, one per row, any order),
// getRows() (row-major uint64, batch elements inside a row) and batch() -- helib_amd::DoubleCRT.
// The wire order is ascending prime index; batch element `b` is the one written.
template <class Poly>
Rows fromPoly(const Poly& poly, size_t phim, int b = 0)
{
  auto idx = poly.getIndexSet();
  std::vector<uint64_t> data = poly.getRows();
  const size_t batch = (size_t)poly.batch(), nr = idx.size();
  if (data.size() != nr * batch * phim || b < 0 || (size_t)b >= batch)
    throw IOError("polynomial shape does not match");
  std::vector<size_t> order(nr);
  for (size_t i = 0; i < nr; i++)
    order[i] = i;
  for (size_t i = 1; i < nr; i++)  // insertion sort by prime index
    for (size_t j = i; j > 0 && idx[order[j]] < idx[order[j - 1]]; j--)
      std::swap(order[j], order[j - 1]);
  Rows r;
  r.n = phim;
  for (size_t k = 0; k < nr; k++) {
    r.idx.push_back((long)idx[order[k]]);
    const uint64_t* src = data.data() + (order[k] * batch + (size_t)b) * phim;
    r.data.insert(r.data.end(), src, src + phim);
  }
  return r;
}
// rows for setRows() of a polynomial whose getIndexSet() is `order` (batch 1)
template <class Idx>
std::vector<uint64_t> toRows(const Rows& r, const Idx& order)
{
  std::vector<uint64_t> out;
  for (auto i : order) {
    size_t k = 0;
    while (k < r.idx.size() && r.idx[k] != (long)i)
      k++;
    if (k == r.idx.size())
      throw IOError("Stream does not contain subset of the context's primes");
    out.insert(out.end(), r.data.begin() + (long)(k * r.n), r.data.begin() + (long)((k + 1) * r.n));
  }
  return out;
}

}  // namespace wire
}  // namespace helib_amd

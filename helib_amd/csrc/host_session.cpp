// host_session.cpp -- the C++17 host (include/helib_amd_ctxt.hpp, helib_amd_keys.hpp) compiled into
// libhelib_amd_host.so behind the C ABI of include/helib_amd_host.h: the reference's benchmark loops
// (benchmarks/bgv_basic.cpp:144-165, benchmarks/ckks_basic.cpp:161-180) with the C++ Ctxt /
// DoubleCRT / SecKey on the timed path.  Host code only: every polynomial operation goes through
// libhelib_amd.so (include/helib_amd.h); there is no CPU arithmetic path here.
#include <cmath>
#include <chrono>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <tuple>
#include <algorithm>
#include <vector>

#include "../../include/helib_amd_host.h"
#include "../../include/helib_amd_keys.hpp"
#include "../../include/helib_amd_io.hpp"

using namespace helib_amd;

static thread_local std::string g_err;
extern "C" const char* hxh_last_error(void) { return g_err.c_str(); }

struct ExportPart;
struct hxh_session {
  int scheme = 0, batch = 1;
  std::unique_ptr<ChainContext> cc;
  std::unique_ptr<Context> dev;
  std::unique_ptr<SecKey> sk;
  std::vector<double> ptxt[2];                 // [b][j]
  std::unique_ptr<Ctxt> fresh[2];              // the two batched operands
  std::unique_ptr<Ctxt> fresh1[2];             // batch element 0 alone
  std::unique_ptr<Ctxt> prod[3];               // kept products: [1] level 1, [2] level 2
  // host copies of a kept product's rows (hxh_decrypt slices elements out of them)
  struct HostParts {
    bool valid = false;
    std::vector<std::pair<SKHandle, std::vector<uint64_t>>> rows;
    std::vector<IndexSet> idx;
  } host[3];
  std::vector<uint64_t> key_blob;              // hxh_export_keys: kept between the size query and the copy
  std::string ct_blob;                         // hxh_export_ctxts: likewise
  std::tuple<int, int, int, int> ct_blob_key{-1, -1, -1, -1};
  // ... and the host copy of the ciphertext it last exported from (dropped by the next multiply of that level)
  std::vector<ExportPart> exp_parts;
  int exp_level = -1, exp_which = -1;
  hxh_session();
  ~hxh_session();
};

template <class F>
static int guarded(F&& f)
{
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  } catch (...) {
    g_err = "unknown exception";
    return -1;
  }
}

static uint64_t sm64(uint64_t& s)
{
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// the part every kind of session shares: chain, device context, arena reservation, key object.
// keys == nullptr: the session makes its own key pair (seed; 0 = OS entropy).  Otherwise the key pair is the
// exported one (SecKey::importKeys; public material only is enough for a session that never decrypts).
// reserve: the multiply loop's working set is taken from the device up front (a session that only holds
// ciphertexts to hand out -- the source of a scatter -- skips it)
static std::unique_ptr<hxh_session> session_base(int device, void* stream, int scheme, long m, long p, long r, long bits,
                                                 int batch, uint64_t seed, const uint64_t* keys, size_t key_words,
                                                 bool reserve = true)
{
  auto s = std::make_unique<hxh_session>();
  s->scheme = scheme;
  s->batch = batch;
  const bool ckks = scheme == 1;
  s->cc = std::make_unique<ChainContext>(m, ckks ? -1 : p, r, bits, 3, 3.2, 10.0, 0, 3, 0, ckks);
  const ChainContext& cc = *s->cc;
  s->dev = cc.makeDeviceContext(device);
  s->dev->setStream(stream);
  if (reserve) {
    // the loop's working set, reserved before anything is timed: about 48 slabs of the largest DoubleCRT of the
    // chain (operands, their mod-switched copies, tensor and key-switch outputs, the kept products of two levels)
    const uint64_t slab = (uint64_t)(cc.ctxtPrimes.size() + cc.specialPrimes.size() + 2) * (uint64_t)batch *
                          (uint64_t)cc.phim * 8u;
    // (not fatal: on a device with less free memory the loop still fits by growing on demand)
    for (uint64_t want = std::min<uint64_t>(48 * slab, (uint64_t)64 << 30); want >= slab; want /= 2) {
      try {
        s->dev->reserve(want);
        break;
      } catch (const std::exception&) {
      }
    }
  }
  s->sk = seed ? std::make_unique<SecKey>(cc, *s->dev, seed) : std::make_unique<SecKey>(cc, *s->dev);
  if (keys)
    s->sk->importKeys(keys, key_words);
  else
    s->sk->GenSecKey(2);   // s^2 -> s: what multiplyBy relinearises with (benchmarks/bgv_basic.cpp:150-152)
  return s;
}

// `batch` ciphertexts given one by one become ONE batched operand: rows packed along the batch axis, the bookkeeping
// of the first element (identical for every element of a fresh batch: it depends on the parameters only), the noise
// estimate the largest of the elements'
struct BatchPacker {
  const ChainContext& cc;
  const Context& dev;
  const KeySet& keys;
  size_t B;
  std::unique_ptr<Ctxt> bt;
  IndexSet idx;
  std::vector<SKHandle> handles;
  std::vector<std::vector<uint64_t>> packed;   // [part] : [row][b][N]
  BatchPacker(const ChainContext& c, const Context& d, const KeySet& k, size_t b) : cc(c), dev(d), keys(k), B(b) {}
  // rows_of(part handle) -> [row][N] of this element, rows in the order of `primes`
  template <class RowsOf>
  void add(size_t b, const PrimeSet& primeSet, long ptxtSpace, long intFactor, double lnNoise, double ptxtMag,
           double lnRatFactor, const std::vector<SKHandle>& hs, const IndexSet& primes, RowsOf&& rows_of)
  {
    const size_t N = (size_t)cc.phim, L = primes.size();
    if (b == 0) {
      bt = std::make_unique<Ctxt>(cc, dev, keys);
      bt->primeSet = primeSet;
      bt->ptxtSpace = ptxtSpace;
      bt->intFactor = intFactor;
      bt->lnNoise = lnNoise;
      bt->ptxtMag = ptxtMag;
      bt->lnRatFactor = lnRatFactor;
      idx = primes;
      handles = hs;
      packed.assign(hs.size(), std::vector<uint64_t>(L * B * N));
    } else {
      if (std::fabs(bt->lnRatFactor - lnRatFactor) > 1e-12 || bt->primeSet != primeSet || bt->intFactor != intFactor ||
          bt->ptxtSpace != ptxtSpace || primes != idx || hs.size() != handles.size())
        throw LogicError("batch elements disagree in their bookkeeping");
      if (lnNoise > (double)bt->lnNoise)
        bt->lnNoise = lnNoise;
      if (ptxtMag > bt->ptxtMag)
        bt->ptxtMag = ptxtMag;
    }
    for (size_t part = 0; part < handles.size(); part++) {
      if (!(hs[part] == handles[part]) || hs[part].powerOfS != handles[part].powerOfS)
        throw LogicError("batch elements disagree in their parts");
      const std::vector<uint64_t> rows = rows_of(part);   // [row][N]
      if (rows.size() != L * N)
        throw LogicError("a ciphertext part has the wrong shape");
      for (size_t row = 0; row < L; row++)
        memcpy(&packed[part][(row * B + b) * N], &rows[row * N], N * 8);
    }
  }
  std::unique_ptr<Ctxt> finish()
  {
    for (size_t part = 0; part < handles.size(); part++) {
      DoubleCRT d(dev, idx, (int)B, DoubleCRT::Uninitialized{});
      d.setRows(packed[part]);
      bt->parts.emplace(handles[part], std::move(d));
      packed[part].clear();
      packed[part].shrink_to_fit();
    }
    return std::move(bt);
  }
};

static int session_create(hxh_session** out, int device, void* stream, int scheme, long m, long p, long r, long bits,
                          int batch, uint64_t seed, const uint64_t* keys, size_t key_words, bool reserve = true)
{
  if (!out || batch < 1 || (scheme != 0 && scheme != 1)) {
    g_err = "hxh_session_create: bad argument";
    return -1;
  }
  return guarded([&] {
    auto s = session_base(device, stream, scheme, m, p, r, bits, batch, seed, keys, key_words, reserve);
    const ChainContext& cc = *s->cc;
    const bool ckks = scheme == 1;
    const size_t N = (size_t)cc.phim, B = (size_t)batch;
    uint64_t ps = seed * 0x9e3779b97f4a7c15ull + 12345;
    // CKKS: the factor PubKey::Encrypt(Ptxt<CKKS>) encodes with, EncryptedArrayCx::encodeScalingFactor() / size with
    // size = 1 (2^11 at m = 65536, precision(1); 2^30 at precision(20))
    const double f = ckks ? (double)cc.encodeScalingFactor() : 1.0;
    // ... and real coefficients uniform in +-1 / (8 sqrt(phi(m)/3)): the canonical embedding of such a polynomial
    // stays below 1 (eight standard deviations of a slot value), the size the encryption declares; slot encoding
    // itself (EncryptedArrayCx::encode) is not on this path
    const double amp = 1.0 / (8.0 * std::sqrt((double)cc.phim / 3.0));
    for (int j = 0; j < 2; j++) {
      s->ptxt[j].resize(B * N);
      BatchPacker pk(cc, *s->dev, s->sk->keys, B);
      for (size_t b = 0; b < B; b++) {
        std::vector<long> msg(N);
        for (size_t i = 0; i < N; i++) {
          if (ckks) {
            const double v = ((double)(sm64(ps) >> 11) / 9007199254740992.0 * 2.0 - 1.0) * amp;
            msg[i] = (long)std::llrint(v * f);
            s->ptxt[j][b * N + i] = (double)msg[i] / f;
          } else {
            msg[i] = (long)(((unsigned __int128)sm64(ps) * (uint64_t)cc.ptxtSpace) >> 64);
            s->ptxt[j][b * N + i] = (double)msg[i];
          }
        }
        Ctxt ct = ckks ? s->sk->CKKSencrypt(msg, 1.0, f) : s->sk->Encrypt(msg);
        if (b == 0)
          s->fresh1[j] = std::make_unique<Ctxt>(ct);
        const std::vector<SKHandle> hs{SKHandle{0, 1}, SKHandle{1, 1}};
        for (auto& h : hs)
          if (ct.parts.at(h).getIndexSet() != cc.ctxtPrimes)
            throw LogicError("a fresh ciphertext is not on the ctxt primes");
        pk.add(b, ct.primeSet, ct.ptxtSpace, ct.intFactor, (double)ct.lnNoise, ct.ptxtMag, ct.lnRatFactor, hs, cc.ctxtPrimes,
               [&](size_t part) { return ct.parts.at(hs[part]).getRows(); });
      }
      s->fresh[j] = pk.finish();
    }
    s->dev->sync();
    *out = s.release();
  });
}

extern "C" int hxh_session_create(hxh_session** out, int device, void* stream, int scheme, long m, long p, long r,
                                  long bits, int batch, uint64_t seed)
{
  return session_create(out, device, stream, scheme, m, p, r, bits, batch, seed, nullptr, 0);
}

extern "C" int hxh_session_create_with_keys(hxh_session** out, int device, void* stream, int scheme, long m, long p,
                                            long r, long bits, int batch, uint64_t enc_seed, const uint64_t* keys,
                                            size_t key_words)
{
  if (!keys || key_words == 0) {
    g_err = "hxh_session_create_with_keys: no key material";
    return -1;
  }
  return session_create(out, device, stream, scheme, m, p, r, bits, batch, enc_seed, keys, key_words);
}

extern "C" int hxh_export_keys(hxh_session* s, uint64_t* out, size_t cap_words, size_t* need_words)
{
  if (!s) {
    g_err = "null session";
    return -1;
  }
  return guarded([&] {
    if (s->key_blob.empty())
      s->key_blob = s->sk->exportKeys();
    if (need_words)
      *need_words = s->key_blob.size();
    if (out) {
      if (cap_words < s->key_blob.size())
        throw InvalidArgument("hxh_export_keys: buffer too small");
      memcpy(out, s->key_blob.data(), s->key_blob.size() * 8);
      s->key_blob.clear();
      s->key_blob.shrink_to_fit();
    }
  });
}

extern "C" int hxh_export_public_keys(hxh_session* s, uint64_t* out, size_t cap_words, size_t* need_words)
{
  if (!s) {
    g_err = "null session";
    return -1;
  }
  return guarded([&] {
    // (not cached: the size query and the copy both build it; the public form is what leaves the process)
    const std::vector<uint64_t> blob = s->sk->exportKeys(false);
    if (need_words)
      *need_words = blob.size();
    if (out) {
      if (cap_words < blob.size())
        throw InvalidArgument("hxh_export_public_keys: buffer too small");
      memcpy(out, blob.data(), blob.size() * 8);
    }
  });
}

static const Ctxt* session_ctxt(const hxh_session* s, int level, int which)
{
  if (level == 0)
    return (which == 0 || which == 1) ? s->fresh[which].get() : nullptr;
  return (level == 1 || level == 2) ? s->prod[level].get() : nullptr;
}

extern "C" int hxh_chain_primes(const hxh_session* s, uint64_t* out, int cap, int* n)
{
  if (!s || !n) {
    g_err = "null argument";
    return -1;
  }
  *n = (int)s->cc->primes.size();
  for (int i = 0; out && i < cap && i < *n; i++)
    out[i] = s->cc->primes[(size_t)i];
  return 0;
}

extern "C" int hxh_ctxt_info(hxh_session* s, int level, int which, double info[8])
{
  if (!s || !info) {
    g_err = "null argument";
    return -1;
  }
  return guarded([&] {
    const Ctxt* c = session_ctxt(s, level, which);
    if (!c)
      throw LogicError("hxh_ctxt_info: no such ciphertext");
    info[0] = (double)c->lnNoise;
    info[1] = c->lnRatFactor;
    info[2] = c->ptxtMag;
    info[3] = (double)c->intFactor;
    info[4] = (double)c->ptxtSpace;
    info[5] = (double)c->parts.size();
    info[6] = s->sk->keys.lnNoise;
    info[7] = (double)(s->sk->keys.ptxtSpace ? s->sk->keys.ptxtSpace : s->cc->ptxtSpace);
  });
}

extern "C" int hxh_ctxt_rows(hxh_session* s, int level, int which, int part, uint64_t* out, int* idx_out, int cap_rows,
                             int* nrows)
{
  if (!s || !nrows || part < 0) {
    g_err = "hxh_ctxt_rows: bad argument";
    return -1;
  }
  return guarded([&] {
    const Ctxt* c = session_ctxt(s, level, which);
    if (!c)
      throw LogicError("hxh_ctxt_rows: no such ciphertext");
    auto it = c->parts.find(SKHandle{(long)part, 1});
    if (it == c->parts.end())
      throw LogicError("hxh_ctxt_rows: no such part");
    const IndexSet idx = it->second.getIndexSet();
    *nrows = (int)idx.size();
    if (idx_out)
      for (int r = 0; r < *nrows && r < cap_rows; r++)
        idx_out[r] = idx[(size_t)r];
    if (out) {
      if (cap_rows < *nrows)
        throw InvalidArgument("hxh_ctxt_rows: buffer too small");
      const std::vector<uint64_t> rows = it->second.getRows();   // [row][batch][phi(m)]
      memcpy(out, rows.data(), rows.size() * 8);
    }
  });
}

// ---- ciphertexts across processes: the batch elements of a session's ciphertext in the reference's binary format
// (Ctxt::writeTo, src/Ctxt.cpp:2584-2611; include/helib_amd_wire.hpp), one after the other ----
struct ExportPart {
  SKHandle h;
  IndexSet idx;
  std::vector<size_t> order;      // rows by ascending prime index (the wire order)
  std::vector<uint64_t> rows;     // [row][B][N], downloaded ONCE per ciphertext (slices of it are exported rank by rank)
};
hxh_session::hxh_session() = default;
hxh_session::~hxh_session() = default;
static std::string export_ctxts(hxh_session* s, const Ctxt& ct, int level, int which, int first, int count)
{
  const size_t N = (size_t)s->cc->phim, B = (size_t)s->batch;
  if (first < 0 || count < 0 || (size_t)first + (size_t)count > B)
    throw InvalidArgument("hxh_export_ctxts: element range outside the batch");
  if (s->exp_level != level || s->exp_which != which) {
    s->exp_parts.clear();
    for (auto& kv : ct.parts) {
      ExportPart x{kv.first, kv.second.getIndexSet(), {}, kv.second.getRows()};
      if (x.rows.size() != x.idx.size() * B * N)
        throw LogicError("hxh_export_ctxts: part of an unexpected shape");
      x.order.resize(x.idx.size());
      for (size_t i = 0; i < x.order.size(); i++)
        x.order[i] = i;
      std::sort(x.order.begin(), x.order.end(), [&](size_t a, size_t b) { return x.idx[a] < x.idx[b]; });
      s->exp_parts.push_back(std::move(x));
    }
    s->exp_level = level;
    s->exp_which = which;
  }
  const std::vector<ExportPart>& hp = s->exp_parts;
  wire::Writer w;
  for (int b = first; b < first + count; b++) {
    wire::CtxtDesc d;
    d.ptxtSpace = ct.ptxtSpace;
    d.intFactor = ct.intFactor;
    d.ptxtMag = wire::xdOf(ct.ptxtMag);
    d.ratFactor = s->cc->ckks ? wire::xdFromLn(ct.lnRatFactor) : wire::XDouble{1.0, 0};
    d.noiseBound = wire::xdFromLn((double)ct.lnNoise);
    d.primeSet.assign(ct.primeSet.begin(), ct.primeSet.end());
    for (auto& x : hp) {
      wire::Part p;
      p.rows.n = N;
      for (size_t k : x.order) {
        p.rows.idx.push_back((long)x.idx[k]);
        const uint64_t* src = x.rows.data() + (k * B + (size_t)b) * N;
        p.rows.data.insert(p.rows.data.end(), src, src + N);
      }
      p.handle[0] = x.h.powerOfS;
      p.handle[1] = x.h.powerOfX;
      p.handle[2] = 0;
      d.parts.push_back(std::move(p));
    }
    wire::write(w, d);
  }
  return w.out;
}

extern "C" int hxh_export_ctxts(hxh_session* s, int level, int which, int first, int count, uint8_t* out, size_t cap_bytes,
                                size_t* need_bytes)
{
  if (!s) {
    g_err = "null session";
    return -1;
  }
  return guarded([&] {
    const Ctxt* c = session_ctxt(s, level, which);
    if (!c)
      throw LogicError("hxh_export_ctxts: no such ciphertext");
    // (the blob is built by the size query and kept until it has been copied out: the rows are downloaded once)
    if (s->ct_blob.empty() || s->ct_blob_key != std::make_tuple(level, which, first, count)) {
      s->ct_blob = export_ctxts(s, *c, level, which, first, count);
      s->ct_blob_key = std::make_tuple(level, which, first, count);
    }
    if (need_bytes)
      *need_bytes = s->ct_blob.size();
    if (out) {
      if (cap_bytes < s->ct_blob.size())
        throw InvalidArgument("hxh_export_ctxts: buffer too small");
      memcpy(out, s->ct_blob.data(), s->ct_blob.size());
      s->ct_blob.clear();
      s->ct_blob.shrink_to_fit();
    }
  });
}

// `batch` wire ciphertexts -> one batched operand.  Ctxt::read's checks (src/Ctxt.cpp:2620-2641, DoubleCRT::read
// :1530-1566): every part on exactly the ciphertext's prime set, primes known to the context, residues below their prime
static std::unique_ptr<Ctxt> import_ctxts(hxh_session* s, const uint8_t* blob, size_t bytes, std::unique_ptr<Ctxt>* first1)
{
  const ChainContext& cc = *s->cc;
  const size_t B = (size_t)s->batch, N = (size_t)cc.phim;
  wire::Reader rd(blob, bytes);
  BatchPacker pk(cc, *s->dev, s->sk->keys, B);
  for (size_t b = 0; b < B; b++) {
    const wire::CtxtDesc d = wire::readCtxt(rd);
    wire::xdCheck(d.ptxtMag, "ptxtMag");
    wire::xdCheck(d.ratFactor, "ratFactor");
    wire::xdCheck(d.noiseBound, "noiseBound");
    PrimeSet ps;
    for (long i : d.primeSet) {
      if (i < 0 || (size_t)i >= cc.primes.size())
        throw wire::IOError("Stream does not contain subset of the context's primes");
      ps.insert((int)i);
    }
    const std::vector<long> want(ps.begin(), ps.end());
    const IndexSet idx(want.begin(), want.end());
    std::vector<SKHandle> hs;
    for (auto& p : d.parts) {
      if (p.rows.idx != want)
        throw wire::IOError("Ciphertext part's index set does not match prime set");
      if (p.rows.n != N)
        throw wire::IOError("Data not valid: d.map[i].length() != phim");
      for (size_t r = 0; r < p.rows.idx.size(); r++) {
        const uint64_t q = cc.primes[(size_t)p.rows.idx[r]];
        for (size_t j = 0; j < N; j++)
          if (p.rows.data[r * N + j] >= q)
            throw wire::IOError("this->map[i][j] invalid: must be between 0 and context.ithPrime(i)");
      }
      hs.push_back(SKHandle{p.handle[0], p.handle[1]});
    }
    pk.add(b, ps, d.ptxtSpace, d.intFactor, wire::lnOf(d.noiseBound), wire::valueOf(d.ptxtMag),
           cc.ckks ? wire::lnOf(d.ratFactor) : 0.0, hs, idx, [&](size_t part) { return d.parts[part].rows.data; });
    if (b == 0 && first1)
      *first1 = std::make_unique<Ctxt>(wire::restore(d, cc, *s->dev, s->sk->keys));
  }
  if (rd.pos != bytes)
    throw wire::IOError("trailing bytes after the last ciphertext");
  return pk.finish();
}

extern "C" int hxh_session_create_from_ctxts(hxh_session** out, int device, void* stream, int scheme, long m, long p, long r,
                                             long bits, int batch, const uint64_t* keys, size_t key_words, const uint8_t* a,
                                             size_t a_bytes, const uint8_t* b, size_t b_bytes)
{
  if (!out || batch < 1 || (scheme != 0 && scheme != 1) || !keys || !key_words || !a || !b) {
    g_err = "hxh_session_create_from_ctxts: bad argument";
    return -1;
  }
  return guarded([&] {
    auto s = session_base(device, stream, scheme, m, p, r, bits, batch, 1, keys, key_words);
    s->fresh[0] = import_ctxts(s.get(), a, a_bytes, &s->fresh1[0]);
    s->fresh[1] = import_ctxts(s.get(), b, b_bytes, &s->fresh1[1]);
    s->dev->sync();
    *out = s.release();
  });
}

extern "C" int hxh_session_create_source(hxh_session** out, int device, void* stream, int scheme, long m, long p, long r,
                                         long bits, int batch, uint64_t seed)
{
  return session_create(out, device, stream, scheme, m, p, r, bits, batch, seed, nullptr, 0, /*reserve=*/false);
}

extern "C" int hxh_decrypt_wire(hxh_session* s, const uint8_t* blob, size_t bytes, double* out, double* bound, size_t* used)
{
  if (!s || !blob || !out) {
    g_err = "hxh_decrypt_wire: bad argument";
    return -1;
  }
  return guarded([&] {
    size_t n_used = 0;
    const Ctxt one = readCtxtFrom(blob, bytes, *s->cc, *s->dev, s->sk->keys, &n_used);
    if (used)
      *used = n_used;
    const size_t N = (size_t)s->cc->phim;
    if (s->scheme == 1) {
      const std::vector<double> v = s->sk->DecryptCKKS(one);   // raw / ratFactor
      memcpy(out, v.data(), N * sizeof(double));
      if (bound)
        *bound = std::exp((double)one.lnNoise - one.lnRatFactor);
    } else {
      const std::vector<long> v = s->sk->Decrypt(one);
      for (size_t i = 0; i < N; i++)
        out[i] = (double)v[i];
      if (bound)
        *bound = one.capacity();
    }
  });
}

extern "C" int hxh_relin_matrix(hxh_session* s, uint64_t* b, uint64_t* a, int* idx_out, int cap_rows, int* ndig,
                                int* nrows)
{
  if (!s || !ndig || !nrows) {
    g_err = "null argument";
    return -1;
  }
  return guarded([&] {
    const KeySwitch* W = s->sk->keys.relin;
    if (!W)
      throw LogicError("hxh_relin_matrix: the session has no relinearisation matrix");
    *ndig = W->ndig();
    *nrows = (int)W->rows().size();
    if (idx_out)
      for (int r = 0; r < *nrows && r < cap_rows; r++)
        idx_out[r] = W->rows()[(size_t)r];
    if (b && a) {
      if (cap_rows < *nrows)
        throw InvalidArgument("hxh_relin_matrix: buffer too small");
      std::vector<uint64_t> hb, ha;
      W->download(hb, ha, (size_t)s->cc->phim);
      memcpy(b, hb.data(), hb.size() * 8);
      memcpy(a, ha.data(), ha.size() * 8);
    }
  });
}

extern "C" int hxh_arena_stats(hxh_session* s, uint64_t out[4])
{
  if (!s || !out) {
    g_err = "null argument";
    return -1;
  }
  return guarded([&] { s->dev->arenaStats(out); });
}

// Batched encryption / decryption timed inside the C++ host (benchmarks/bgv_basic.cpp:186-211 time ONE
// PubKey::Encrypt / SecKey::Decrypt per iteration): SecKey::EncryptBatch / DecryptBatch over `batch` random plaintexts,
// `reps` times each, wall clock with the device drained.  out = {ms per ciphertext encrypted, ms per ciphertext
// decrypted, batch, 1.0 when every element decrypted to its plaintext}.  BGV only.
extern "C" int hxh_encrypt_decrypt_batch(hxh_session* s, int batch, int reps, double out[4])
{
  if (!s || !out || batch < 1 || reps < 1) {
    g_err = "bad argument";
    return -1;
  }
  return guarded([&] {
    const ChainContext& cc = *s->cc;
    if (cc.ckks)
      throw LogicError("hxh_encrypt_decrypt_batch: BGV sessions only");
    const size_t n = (size_t)cc.phim;
    std::vector<long> msgs((size_t)batch * n);
    uint64_t ps = 0x243f6a8885a308d3ull;
    for (auto& v : msgs)
      v = (long)(((unsigned __int128)sm64(ps) * (uint64_t)cc.ptxtSpace) >> 64);
    Ctxt warm = s->sk->EncryptBatch(msgs, batch);   // (plans, arena slabs)
    s->dev->sync();
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) {
      Ctxt ct = s->sk->EncryptBatch(msgs, batch);
      s->dev->sync();
    }
    auto t1 = std::chrono::steady_clock::now();
    std::vector<long> dec;
    for (int r = 0; r < reps; r++)
      dec = s->sk->DecryptBatch(warm);
    auto t2 = std::chrono::steady_clock::now();
    out[0] = std::chrono::duration<double, std::milli>(t1 - t0).count() / ((double)reps * batch);
    out[1] = std::chrono::duration<double, std::milli>(t2 - t1).count() / ((double)reps * batch);
    out[2] = (double)batch;
    out[3] = dec == msgs ? 1.0 : 0.0;
  });
}

extern "C" int hxh_session_destroy(hxh_session* s)
{
  return guarded([&] { delete s; });
}

extern "C" int hxh_session_info(const hxh_session* s, long info[8])
{
  if (!s || !info) {
    g_err = "null argument";
    return -1;
  }
  auto bitsOf = [](uint64_t q) {
    long b = 0;
    while (q) {
      b++;
      q >>= 1;
    }
    return b;
  };
  const ChainContext& cc = *s->cc;
  info[0] = cc.phim;
  info[1] = (long)cc.ctxtPrimes.size();
  info[2] = (long)cc.specialPrimes.size();
  info[3] = (long)cc.digits.size();
  info[4] = (long)cc.smallPrimes.size();
  info[5] = bitsOf(cc.primes[(size_t)cc.ctxtPrimes[0]]);
  info[6] = cc.specialPrimes.empty() ? 0 : bitsOf(cc.primes[(size_t)cc.specialPrimes[0]]);
  info[7] = s->batch;
  return 0;
}

static void run_loop(hxh_session* s, const Ctxt& a0, const Ctxt& b0, int k, int measure, std::unique_ptr<Ctxt>& keep)
{
  Ctxt::deferNorms() = measure != 0;
  std::unique_ptr<Ctxt> prev;
  for (int i = 0; i < k; i++) {
    auto a = std::make_unique<Ctxt>(a0);   // copy(ctxt1): benchmarks/bgv_basic.cpp:160 (copy-on-write on the device)
    a->measure = measure != 0;
    Ctxt b = b0;                            // Ctxt::multLowLvl's own copy of `other` (src/Ctxt.cpp:1716-1745)
    b.measure = measure != 0;
    a->multiplyBy(std::move(b));
    if (prev)
      (void)(double)prev->lnNoise;          // the previous result's estimate, one multiply later
    prev = std::move(a);                    // ... and the previous result is dropped (its slabs recycle)
  }
  if (prev)
    (void)(double)prev->lnNoise;
  keep = std::move(prev);
}

extern "C" int hxh_multiply(hxh_session* s, int level, int k, int measure)
{
  if (!s || k < 1 || (level != 1 && level != 2)) {
    g_err = "hxh_multiply: bad argument";
    return -1;
  }
  return guarded([&] {
    if (level == 1) {
      run_loop(s, *s->fresh[0], *s->fresh[1], k, measure, s->prod[1]);
      s->host[1].valid = false;
      if (s->exp_level == 1)
        s->exp_level = -1;
    } else {
      if (!s->prod[1])
        throw LogicError("hxh_multiply: level 2 needs a level-1 product (call level 1 first)");
      run_loop(s, *s->prod[1], *s->prod[1], k, measure, s->prod[2]);
      s->host[2].valid = false;
      if (s->exp_level == 2)
        s->exp_level = -1;
    }
  });
}

extern "C" int hxh_multiply_single(hxh_session* s, int measure)
{
  if (!s) {
    g_err = "null session";
    return -1;
  }
  return guarded([&] {
    std::unique_ptr<Ctxt> keep;
    run_loop(s, *s->fresh1[0], *s->fresh1[1], 1, measure, keep);
  });
}

extern "C" int hxh_plaintext(const hxh_session* s, int which, double* out)
{
  if (!s || !out || which < 0 || which > 1) {
    g_err = "hxh_plaintext: bad argument";
    return -1;
  }
  if (s->ptxt[which].empty()) {
    g_err = "hxh_plaintext: this session holds ciphertexts it was handed (hxh_session_create_from_ctxts): no plaintexts";
    return -1;
  }
  memcpy(out, s->ptxt[which].data(), s->ptxt[which].size() * sizeof(double));
  return 0;
}

extern "C" int hxh_decrypt(hxh_session* s, int level, int b, double* out, double* bound)
{
  if (!s || !out || level < 0 || level > 2 || b < 0 || b >= s->batch) {
    g_err = "hxh_decrypt: bad argument";
    return -1;
  }
  return guarded([&] {
    const Ctxt* src = level == 0 ? s->fresh[0].get() : s->prod[level].get();
    if (!src)
      throw LogicError("hxh_decrypt: no product kept for this level");
    hxh_session::HostParts& H = s->host[level];
    const size_t N = (size_t)s->cc->phim, B = (size_t)s->batch;
    if (!H.valid) {
      H.rows.clear();
      H.idx.clear();
      for (auto& kv : src->parts) {
        H.rows.emplace_back(kv.first, kv.second.getRows());
        H.idx.push_back(kv.second.getIndexSet());
      }
      H.valid = true;
    }
    Ctxt one(*s->cc, *s->dev, s->sk->keys);
    one.primeSet = src->primeSet;
    one.ptxtSpace = src->ptxtSpace;
    one.intFactor = src->intFactor;
    one.lnNoise = (double)src->lnNoise;
    one.ptxtMag = src->ptxtMag;
    one.lnRatFactor = src->lnRatFactor;
    for (size_t i = 0; i < H.rows.size(); i++) {
      const size_t R = H.idx[i].size();
      std::vector<uint64_t> rows(R * N);
      for (size_t r = 0; r < R; r++)
        memcpy(&rows[r * N], &H.rows[i].second[(r * B + (size_t)b) * N], N * 8);
      DoubleCRT d(*s->dev, H.idx[i], 1, DoubleCRT::Uninitialized{});
      d.setRows(rows);
      one.parts.emplace(H.rows[i].first, std::move(d));
    }
    if (s->scheme == 1) {
      const std::vector<double> v = s->sk->DecryptCKKS(one);   // raw / ratFactor
      memcpy(out, v.data(), N * sizeof(double));
      if (bound)
        *bound = std::exp((double)src->lnNoise - src->lnRatFactor);
    } else {
      const std::vector<long> v = s->sk->Decrypt(one);
      for (size_t i = 0; i < N; i++)
        out[i] = (double)v[i];
      if (bound)
        *bound = src->capacity();
    }
  });
}

extern "C" int hxh_result_primes(const hxh_session* s, int level, int* out, int cap, int* n)
{
  if (!s || !n || level < 0 || level > 2) {
    g_err = "hxh_result_primes: bad argument";
    return -1;
  }
  const Ctxt* src = level == 0 ? s->fresh[0].get() : s->prod[level].get();
  if (!src) {
    g_err = "no product kept for this level";
    return -1;
  }
  *n = (int)src->primeSet.size();
  int i = 0;
  for (int pidx : src->primeSet)
    if (out && i < cap)
      out[i++] = pidx;
  return 0;
}

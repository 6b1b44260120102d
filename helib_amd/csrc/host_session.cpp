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
#include <vector>

#include "../../include/helib_amd_host.h"
#include "../../include/helib_amd_keys.hpp"

using namespace helib_amd;

static thread_local std::string g_err;
extern "C" const char* hxh_last_error(void) { return g_err.c_str(); }

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

// keys == nullptr: the session makes its own key pair (seed; 0 = OS entropy).  Otherwise the key pair is the
// exported one (SecKey::importKeys) and `seed` drives this session's encryption randomness and plaintexts only.
static int session_create(hxh_session** out, int device, void* stream, int scheme, long m, long p, long r, long bits,
                          int batch, uint64_t seed, const uint64_t* keys, size_t key_words)
{
  if (!out || batch < 1 || (scheme != 0 && scheme != 1)) {
    g_err = "hxh_session_create: bad argument";
    return -1;
  }
  return guarded([&] {
    auto s = std::make_unique<hxh_session>();
    s->scheme = scheme;
    s->batch = batch;
    const bool ckks = scheme == 1;
    s->cc = std::make_unique<ChainContext>(m, ckks ? -1 : p, r, bits, 3, 3.2, 10.0, 0, 3, 0, ckks);
    const ChainContext& cc = *s->cc;
    s->dev = cc.makeDeviceContext(device);
    s->dev->setStream(stream);
    {
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
    const size_t N = (size_t)cc.phim, L = cc.ctxtPrimes.size(), B = (size_t)batch;
    uint64_t ps = seed * 0x9e3779b97f4a7c15ull + 12345;
    // CKKS: the factor PubKey::Encrypt(Ptxt<CKKS>) encodes with, EncryptedArrayCx::encodeScalingFactor() / size with
    // size = 1 (2^11 at m = 65536, precision(1); 2^30 at precision(20))
    const double f = ckks ? (double)cc.encodeScalingFactor() : 1.0;
    // ... and real coefficients uniform in +-1 / (8 sqrt(phi(m)/3)): the canonical embedding of such a polynomial
    // stays below 1 (eight standard deviations of a slot value), the size the encryption declares; slot encoding
    // itself (EncryptedArrayCx::encode) is not on this path
    const double amp = 1.0 / (8.0 * std::sqrt((double)cc.phim / 3.0));
    std::vector<uint64_t> packed[2][2];   // [operand][part] : [row][b][N]
    for (int j = 0; j < 2; j++) {
      s->ptxt[j].resize(B * N);
      for (int part = 0; part < 2; part++)
        packed[j][part].resize(L * B * N);
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
        const SKHandle h[2] = {SKHandle{0, 1}, SKHandle{1, 1}};
        for (int part = 0; part < 2; part++) {
          const DoubleCRT& d = ct.parts.at(h[part]);
          if (d.getIndexSet() != cc.ctxtPrimes)
            throw LogicError("a fresh ciphertext is not on the ctxt primes");
          const std::vector<uint64_t> rows = d.getRows();   // [row][1][N]
          for (size_t row = 0; row < L; row++)
            memcpy(&packed[j][part][(row * B + b) * N], &rows[row * N], N * 8);
        }
        if (b == 0) {
          // the batched operand carries the bookkeeping of its first element (identical for every
          // element: it depends on the parameters only, not on the random draws -- checked below)
          s->fresh[j] = std::make_unique<Ctxt>(cc, *s->dev, s->sk->keys);
          Ctxt& bt = *s->fresh[j];
          bt.primeSet = ct.primeSet;
          bt.ptxtSpace = ct.ptxtSpace;
          bt.intFactor = ct.intFactor;
          bt.lnNoise = (double)ct.lnNoise;
          bt.ptxtMag = ct.ptxtMag;
          bt.lnRatFactor = ct.lnRatFactor;
        } else {
          const Ctxt& bt = *s->fresh[j];
          if (std::fabs(bt.lnRatFactor - ct.lnRatFactor) > 1e-12 || bt.primeSet != ct.primeSet ||
              bt.intFactor != ct.intFactor)
            throw LogicError("batch elements disagree in their bookkeeping");
          // (a batched ciphertext's noise estimate is the largest of its elements')
          if ((double)ct.lnNoise > (double)bt.lnNoise)
            s->fresh[j]->lnNoise = (double)ct.lnNoise;
        }
      }
      for (int part = 0; part < 2; part++) {
        DoubleCRT d(*s->dev, cc.ctxtPrimes, batch, DoubleCRT::Uninitialized{});
        d.setRows(packed[j][part]);
        s->fresh[j]->parts.emplace(SKHandle{(long)part, 1}, std::move(d));
        packed[j][part].clear();
        packed[j][part].shrink_to_fit();
      }
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
    } else {
      if (!s->prod[1])
        throw LogicError("hxh_multiply: level 2 needs a level-1 product (call level 1 first)");
      run_loop(s, *s->prod[1], *s->prod[1], k, measure, s->prod[2]);
      s->host[2].valid = false;
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

// hx_mock.cpp -- TEST INFRASTRUCTURE.  The entry points of include/helib_amd.h that the C++ host headers
// (helib_amd.hpp, helib_amd_ctxt.hpp, helib_amd_keys.hpp) call, implemented on the CPU over the oracle
// (oracle/hx_oracle.c), so that the host-side control flow of the C++ headers -- prime-set decisions,
// noise bookkeeping, handle algebra, key management -- can be exercised by the `-m "not gpu"` suite
// with C++ programs that decrypt what they computed.  It is built into tests/cpp/libhx_mock.so by the
// tests themselves and linked ONLY by them; the product library (helib_amd/lib/libhelib_amd.so) has no
// CPU path and nothing in the package refers to this file.  Layout and semantics as helib_amd.h states
// them: a poly is [row][batch][phi(m)], rows are addressed by prime index.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "helib_amd.h"
#include "hx_oracle.h"

struct hx_ctx {
  ho_ctx* o = nullptr;
  uint64_t m = 0;
  long phim = 0;
  std::vector<uint64_t> q, root;
  // deferred read-back of norms, as the engine does it: with hx_ctx_defer_norms(ctx, 1) a *_norms call leaves the
  // caller's array untouched (here: poisoned with NaN) until hx_norms_flush writes the values -- a host side that
  // reads too early, or frees the array before the flush, shows up in the CPU tests (the latter under ASan)
  bool defer = false;
  std::vector<std::pair<double*, std::vector<double>>> pending;
};
static void deliver(hx_ctx* c, double* out, std::vector<double> vals)
{
  if (!out)
    return;
  if (!c->defer) {
    std::copy(vals.begin(), vals.end(), out);
    return;
  }
  for (size_t i = 0; i < vals.size(); i++)
    out[i] = std::nan("");
  c->pending.emplace_back(out, std::move(vals));
}
struct hx_poly {
  hx_ctx* ctx;
  int batch;
  std::vector<int> idx;
  std::vector<uint64_t> d;  // [row][batch][N]
  size_t N() const { return (size_t)ctx->phim; }
  size_t rw() const { return (size_t)batch * N(); }
  int nrows() const { return (int)idx.size(); }
  uint64_t* row(int r, int b) { return d.data() + ((size_t)r * batch + b) * N(); }
  const uint64_t* row(int r, int b) const { return d.data() + ((size_t)r * batch + b) * N(); }
  // rows of one batch element, contiguous [row][N]
  std::vector<uint64_t> elem(int b) const
  {
    std::vector<uint64_t> out((size_t)nrows() * N());
    for (int r = 0; r < nrows(); r++)
      memcpy(out.data() + (size_t)r * N(), row(r, b), N() * 8);
    return out;
  }
  void put(int b, const std::vector<uint64_t>& rows)
  {
    for (int r = 0; r < nrows(); r++)
      memcpy(row(r, b), rows.data() + (size_t)r * N(), N() * 8);
  }
  void reshape(const std::vector<int>& s)
  {
    idx = s;
    d.assign((size_t)s.size() * rw(), 0);
  }
};
struct hx_ksk {
  hx_ctx* ctx;
  int ndig;
  std::vector<int> rows;
  std::vector<uint64_t> b, a;  // [ndig][nrows][N]
};
struct hx_graph {
  int unused;
};

static thread_local std::string g_err;
static int fail(int code, const std::string& msg)
{
  g_err = msg;
  return code;
}
static int find(const std::vector<int>& v, int x)
{
  auto it = std::find(v.begin(), v.end(), x);
  return it == v.end() ? -1 : (int)(it - v.begin());
}

extern "C" {

const char* hx_last_error(void) { return g_err.c_str(); }
const char* hx_version(void) { return "helib_amd mock over the CPU oracle (tests only)"; }
int hx_device_count(int* count)
{
  *count = 0;
  return HX_OK;
}

int hx_ctx_create(hx_ctx** out, int, uint64_t m)
{
  hx_ctx* c = new hx_ctx();
  c->o = ho_ctx_create(m);
  c->m = m;
  c->phim = ho_ctx_phim(c->o);
  *out = c;
  return HX_OK;
}
int hx_ctx_destroy(hx_ctx* c)
{
  if (c) {
    ho_ctx_destroy(c->o);
    delete c;
  }
  return HX_OK;
}
int hx_ctx_phim(const hx_ctx* c, uint64_t* phim)
{
  *phim = (uint64_t)c->phim;
  return HX_OK;
}
int hx_ctx_set_stream(hx_ctx*, void*) { return HX_OK; }
int hx_ctx_reserve(hx_ctx*, uint64_t) { return HX_OK; }
int hx_ctx_arena_stats(hx_ctx*, uint64_t out[4])
{
  out[0] = out[1] = out[2] = out[3] = 0;
  return HX_OK;
}
int hx_ctx_sync(hx_ctx*) { return HX_OK; }
int hx_ctx_add_prime(hx_ctx* c, uint64_t q, uint64_t root, int* idx_out)
{
  if (q >> 60)
    return fail(HX_ERR_INVALID, "prime must be below 2^60");
  int i = ho_ctx_add_prime(c->o, q, root);
  c->q.push_back(q);
  c->root.push_back(ho_ctx_root(c->o, i));
  if (idx_out)
    *idx_out = i;
  return HX_OK;
}
int hx_ctx_num_primes(const hx_ctx* c, int* n)
{
  *n = (int)c->q.size();
  return HX_OK;
}
int hx_ctx_prime(const hx_ctx* c, int idx, uint64_t* q, uint64_t* root)
{
  if (idx < 0 || idx >= (int)c->q.size())
    return fail(HX_ERR_INVALID, "prime index out of range");
  if (q)
    *q = c->q[(size_t)idx];
  if (root)
    *root = c->root[(size_t)idx];
  return HX_OK;
}

int hx_poly_create(hx_ctx* ctx, int batch, const int* prime_idx, int nrows, hx_poly** out)
{
  if (!ctx || batch < 1 || nrows < 0)
    return fail(HX_ERR_INVALID, "bad poly shape");
  for (int r = 0; r < nrows; r++)
    if (prime_idx[r] < 0 || prime_idx[r] >= (int)ctx->q.size())
      return fail(HX_ERR_INVALID, "prime index out of range");
  hx_poly* p = new hx_poly{ctx, batch, {}, {}};
  p->reshape(std::vector<int>(prime_idx, prime_idx + nrows));
  *out = p;
  return HX_OK;
}
int hx_poly_create_uninit(hx_ctx* ctx, int batch, const int* prime_idx, int nrows, hx_poly** out)
{
  return hx_poly_create(ctx, batch, prime_idx, nrows, out);
}
int hx_poly_destroy(hx_poly* p)
{
  delete p;
  return HX_OK;
}
int hx_poly_shape(const hx_poly* p, int* batch, int* nrows, uint64_t* phim)
{
  if (batch)
    *batch = p->batch;
  if (nrows)
    *nrows = p->nrows();
  if (phim)
    *phim = (uint64_t)p->ctx->phim;
  return HX_OK;
}
int hx_poly_primes(const hx_poly* p, int* out)
{
  std::copy(p->idx.begin(), p->idx.end(), out);
  return HX_OK;
}
int hx_poly_upload(hx_poly* p, const uint64_t* host)
{
  memcpy(p->d.data(), host, p->d.size() * 8);
  return HX_OK;
}
int hx_poly_download(const hx_poly* p, uint64_t* host)
{
  memcpy(host, p->d.data(), p->d.size() * 8);
  return HX_OK;
}
int hx_poly_copy(hx_poly* dst, const hx_poly* src)
{
  if (!dst || !src || dst->ctx != src->ctx || dst->batch != src->batch)
    return fail(HX_ERR_INVALID, "Context mismatch");
  if (dst != src) {
    dst->idx = src->idx;
    dst->d = src->d;
  }
  return HX_OK;
}
int hx_poly_set_zero(hx_poly* p)
{
  std::fill(p->d.begin(), p->d.end(), 0);
  return HX_OK;
}
int hx_randomize(hx_poly* p, const uint8_t* key32, uint64_t stream)
{
  uint32_t key[8];
  memcpy(key, key32, 32);
  for (int r = 0; r < p->nrows(); r++)
    for (int b = 0; b < p->batch; b++) {
      uint32_t nonce[3] = {(uint32_t)stream, (uint32_t)(stream >> 32),
                           ((uint32_t)p->idx[(size_t)r] & 0xffffu) | (((uint32_t)b & 0xffffu) << 16)};
      ho_randomize_row(p->row(r, b), p->ctx->phim, p->ctx->q[(size_t)p->idx[(size_t)r]], key, nonce);
    }
  return HX_OK;
}
int hx_poly_remove_primes(hx_poly* p, const int* prime_idx, int n)
{
  std::vector<int> keep;
  std::vector<uint64_t> nd;
  for (int r = 0; r < p->nrows(); r++) {
    if (std::find(prime_idx, prime_idx + n, p->idx[(size_t)r]) != prime_idx + n)
      continue;
    keep.push_back(p->idx[(size_t)r]);
    nd.insert(nd.end(), p->row(r, 0), p->row(r, 0) + p->rw());
  }
  p->idx = keep;
  p->d = nd;
  return HX_OK;
}

static int transform(hx_poly* p, bool fwd)
{
  for (int b = 0; b < p->batch; b++) {
    std::vector<uint64_t> in = p->elem(b), out(in.size());
    if (fwd)
      ho_dcrt_fft(p->ctx->o, p->idx.data(), p->nrows(), in.data(), out.data());
    else
      ho_dcrt_ifft(p->ctx->o, p->idx.data(), p->nrows(), in.data(), out.data());
    p->put(b, out);
  }
  return HX_OK;
}
int hx_ntt_forward(hx_poly* p) { return transform(p, true); }
int hx_ntt_inverse(hx_poly* p) { return transform(p, false); }

static int binary(hx_poly* a, const hx_poly* b, int op)
{
  if (a->ctx != b->ctx)
    return fail(HX_ERR_INVALID, "Context mismatch");
  if (b->batch != a->batch && b->batch != 1)      // (as the engine: a batch-1 operand is broadcast over the batch)
    return fail(HX_ERR_INVALID, "batch mismatch");
  for (int r = 0; r < a->nrows(); r++)
    if (find(b->idx, a->idx[(size_t)r]) < 0)
      return fail(HX_ERR_PRIMESET, "DoubleCRT::Op: the prime set of the operand does not cover this object's");
  const long n1 = (long)a->rw() / a->batch;       // one element of a row
  for (int r = 0; r < a->nrows(); r++) {
    const int rb = find(b->idx, a->idx[(size_t)r]);
    const uint64_t q = a->ctx->q[(size_t)a->idx[(size_t)r]];
    for (int e = 0; e < a->batch; e++) {
      uint64_t* x = a->row(r, 0) + (size_t)e * (size_t)n1;
      const uint64_t* y = b->row(rb, 0) + (b->batch == 1 ? 0 : (size_t)e * (size_t)n1);
      if (op == 0)
        ho_row_add(x, x, y, n1, q);
      else if (op == 1)
        ho_row_sub(x, x, y, n1, q);
      else
        ho_row_mul(x, x, y, n1, q);
    }
  }
  return HX_OK;
}
int hx_add(hx_poly* a, const hx_poly* b) { return binary(a, b, 0); }
int hx_sub(hx_poly* a, const hx_poly* b) { return binary(a, b, 1); }
int hx_mul(hx_poly* a, const hx_poly* b) { return binary(a, b, 2); }
int hx_negate(hx_poly* a)
{
  for (int r = 0; r < a->nrows(); r++)
    ho_row_neg(a->row(r, 0), a->row(r, 0), (long)a->rw(), a->ctx->q[(size_t)a->idx[(size_t)r]]);
  return HX_OK;
}
static int scalar(hx_poly* a, const uint64_t* c, int op)
{
  for (int r = 0; r < a->nrows(); r++) {
    const uint64_t q = a->ctx->q[(size_t)a->idx[(size_t)r]];
    uint64_t* x = a->row(r, 0);
    const long n = (long)a->rw();
    if (op == 0)
      ho_row_add_scalar(x, x, c[r] % q, n, q);
    else if (op == 1)
      ho_row_sub_scalar(x, x, c[r] % q, n, q);
    else if (op == 2)
      ho_row_mul_scalar(x, x, c[r] % q, n, q);
    else
      std::fill(x, x + n, c[r] % q);
  }
  return HX_OK;
}
int hx_add_scalar(hx_poly* a, const uint64_t* c) { return scalar(a, c, 0); }
int hx_sub_scalar(hx_poly* a, const uint64_t* c) { return scalar(a, c, 1); }
int hx_mul_scalar(hx_poly* a, const uint64_t* c) { return scalar(a, c, 2); }
int hx_set_scalar(hx_poly* a, const uint64_t* c) { return scalar(a, c, 3); }
int hx_exp(hx_poly* a, uint64_t e)
{
  for (int r = 0; r < a->nrows(); r++) {
    const uint64_t q = a->ctx->q[(size_t)a->idx[(size_t)r]];
    uint64_t* x = a->row(r, 0);
    for (size_t j = 0; j < a->rw(); j++)
      x[j] = ho_powmod(x[j], e, q);
  }
  return HX_OK;
}
int hx_automorph(hx_poly* a, uint64_t k)
{
  const uint64_t m = a->ctx->m;
  k %= m;
  std::vector<uint64_t> out(a->N());
  for (int r = 0; r < a->nrows(); r++)
    for (int b = 0; b < a->batch; b++) {
      if (ho_row_automorph(out.data(), a->row(r, b), m, ho_ctx_zms(a->ctx->o), a->ctx->phim, k) != 0)
        return fail(HX_ERR_NOT_IN_ZMSTAR, "DoubleCRT::automorph: k not in Zm*");
      memcpy(a->row(r, b), out.data(), a->N() * 8);
    }
  return HX_OK;
}
int hx_complex_conj(hx_poly* a) { return hx_automorph(a, a->ctx->m - 1); }

int hx_add_primes_and_scale(hx_poly* a, const int* add_idx, int nadd)
{
  for (int i = 0; i < nadd; i++)
    if (find(a->idx, add_idx[i]) >= 0)
      return fail(HX_ERR_PRIMESET, "addPrimesAndScale: prime already present");
  std::vector<int> nidx = a->idx;
  nidx.insert(nidx.end(), add_idx, add_idx + nadd);
  hx_poly t{a->ctx, a->batch, {}, {}};
  t.reshape(nidx);
  for (int b = 0; b < a->batch; b++) {
    std::vector<uint64_t> rows = a->elem(b);
    ho_dcrt_scale_by_primes(a->ctx->o, a->idx.data(), a->nrows(), rows.data(), add_idx, nadd);
    rows.resize(nidx.size() * a->N(), 0);
    t.put(b, rows);
  }
  a->idx = t.idx;
  a->d = t.d;
  return HX_OK;
}
int hx_add_primes(hx_poly* a, const int* add_idx, int nadd)
{
  std::vector<int> nidx = a->idx;
  nidx.insert(nidx.end(), add_idx, add_idx + nadd);
  hx_poly t{a->ctx, a->batch, {}, {}};
  t.reshape(nidx);
  for (int b = 0; b < a->batch; b++) {
    std::vector<uint64_t> rows = a->elem(b), ext((size_t)nadd * a->N());
    ho_dcrt_add_primes(a->ctx->o, a->idx.data(), a->nrows(), rows.data(), add_idx, nadd, ext.data(), nullptr);
    rows.insert(rows.end(), ext.begin(), ext.end());
    t.put(b, rows);
  }
  a->idx = t.idx;
  a->d = t.d;
  return HX_OK;
}
int hx_poly_rem(const hx_poly* a, uint64_t t, uint64_t* out_host)
{
  if (t < 2)
    return fail(HX_ERR_INVALID, "modulus must be at least 2");
  int bits = 0;
  for (int i : a->idx)
    bits += 64 - __builtin_clzll(a->ctx->q[(size_t)i]);
  const int nl = bits / 64 + 3;
  std::vector<uint64_t> mag(a->N() * (size_t)nl);
  std::vector<int8_t> sign(a->N());
  for (int b = 0; b < a->batch; b++) {
    std::vector<uint64_t> rows = a->elem(b);
    if (ho_dcrt_to_poly_limbs(a->ctx->o, a->idx.data(), a->nrows(), rows.data(), 0, mag.data(), nl, sign.data()) != 0)
      return fail(HX_ERR_UNSUPPORTED, "toPoly: too many primes for the mock");
    for (size_t j = 0; j < a->N(); j++) {
      unsigned __int128 r = 0;
      for (int l = nl - 1; l >= 0; l--)
        r = ((r << 64) | mag[j * (size_t)nl + (size_t)l]) % t;
      uint64_t v = (uint64_t)r;
      if (sign[j] < 0 && v)
        v = t - v;
      out_host[(size_t)b * a->N() + j] = v;
    }
  }
  return HX_OK;
}

// scale_down of every batch element; norms (optional): embeddingLargestCoeff(fdelta) per element
static int scale_down_one(hx_poly* a, const int* drop, int ndrop, uint64_t ptxt, double* norms, double* fdelta)
{
  std::vector<int> keep;
  for (int i : a->idx)
    if (std::find(drop, drop + ndrop, i) == drop + ndrop)
      keep.push_back(i);
  for (int i = 0; i < ndrop; i++)
    if (find(a->idx, drop[i]) < 0)
      return fail(HX_ERR_PRIMESET, "scaleDownToSet: dropped prime not present");
  if (keep.empty())
    return fail(HX_ERR_PRIMESET, "scaleDownToSet: nothing would be left");
  hx_poly t{a->ctx, a->batch, {}, {}};
  t.reshape(keep);
  std::vector<double> fd(a->N()), nv(norms ? (size_t)a->batch : 0);
  for (int b = 0; b < a->batch; b++) {
    std::vector<uint64_t> rows = a->elem(b), out(keep.size() * a->N());
    ho_dcrt_scale_down(a->ctx->o, a->idx.data(), a->nrows(), rows.data(), drop, ndrop, ptxt ? ptxt : 1, out.data(),
                       (norms || fdelta) ? fd.data() : nullptr);
    t.put(b, out);
    if (norms)
      nv[(size_t)b] = ho_embedding_largest_coeff(a->ctx->m, fd.data(), a->ctx->phim);
    if (fdelta)
      memcpy(fdelta + (size_t)b * a->N(), fd.data(), a->N() * 8);
  }
  a->idx = t.idx;
  a->d = t.d;
  deliver(a->ctx, norms, std::move(nv));
  return HX_OK;
}
int hx_scale_down(hx_poly* a, const int* drop_idx, int ndrop, uint64_t ptxt_space)
{
  if (ndrop == 0)
    return HX_OK;
  return scale_down_one(a, drop_idx, ndrop, ptxt_space, nullptr, nullptr);
}
int hx_scale_down_multi_norms(hx_poly** polys, int npoly, const int* drop_idx, int ndrop, uint64_t ptxt_space,
                              double* norms, double* fdelta)
{
  for (int i = 0; i < npoly; i++) {
    hx_poly* a = polys[i];
    int rc = scale_down_one(a, drop_idx, ndrop, ptxt_space, norms ? norms + (size_t)i * a->batch : nullptr,
                            fdelta ? fdelta + (size_t)i * a->batch * a->N() : nullptr);
    if (rc)
      return rc;
  }
  return HX_OK;
}
int hx_scale_down_multi(hx_poly** polys, int npoly, const int* drop_idx, int ndrop, uint64_t ptxt_space)
{
  return hx_scale_down_multi_norms(polys, npoly, drop_idx, ndrop, ptxt_space, nullptr, nullptr);
}
int hx_bring_to_set_multi_norms(hx_poly** polys, int npoly, const int* add_idx, int nadd, const int* drop_idx,
                                int ndrop, uint64_t ptxt_space, double* norms)
{
  for (int i = 0; i < npoly; i++) {
    int rc = nadd ? hx_add_primes_and_scale(polys[i], add_idx, nadd) : HX_OK;
    if (rc)
      return rc;
  }
  if (ndrop == 0)
    return HX_OK;
  return hx_scale_down_multi_norms(polys, npoly, drop_idx, ndrop, ptxt_space, norms, nullptr);
}
int hx_bring_to_set_multi(hx_poly** polys, int npoly, const int* add_idx, int nadd, const int* drop_idx, int ndrop,
                          uint64_t ptxt_space)
{
  return hx_bring_to_set_multi_norms(polys, npoly, add_idx, nadd, drop_idx, ndrop, ptxt_space, nullptr);
}

int hx_break_into_digits_norms(const hx_poly* a, const int* dig_idx, const int* dig_off, int ndig, const int* sp_idx,
                               int nsp, hx_poly* out, double* norms)
{
  std::vector<int> allp = a->idx;
  allp.insert(allp.end(), sp_idx, sp_idx + nsp);
  std::vector<int> oidx;
  for (int d = 0; d < ndig; d++)
    oidx.insert(oidx.end(), allp.begin(), allp.end());
  out->batch = a->batch;
  out->reshape(oidx);
  std::vector<double> nrm((size_t)ndig), nv(norms ? (size_t)ndig * (size_t)a->batch : 0);
  for (int b = 0; b < a->batch; b++) {
    std::vector<uint64_t> rows = a->elem(b), dg(oidx.size() * a->N());
    ho_dcrt_break_into_digits_norms(a->ctx->o, a->idx.data(), a->nrows(), rows.data(), dig_idx, dig_off, ndig,
                                    allp.data(), (int)allp.size(), dg.data(), norms ? nrm.data() : nullptr);
    out->put(b, dg);
    if (norms)
      for (int d = 0; d < ndig; d++)
        nv[(size_t)d * a->batch + (size_t)b] = nrm[(size_t)d];
  }
  deliver(a->ctx, norms, std::move(nv));
  return HX_OK;
}
int hx_break_into_digits(const hx_poly* a, const int* dig_idx, const int* dig_off, int ndig, const int* sp_idx, int nsp,
                         hx_poly* out)
{
  return hx_break_into_digits_norms(a, dig_idx, dig_off, ndig, sp_idx, nsp, out, nullptr);
}

int hx_ksk_create(hx_ctx* ctx, int ndig, const int* row_idx, int nrows, const uint64_t* b, const uint64_t* a,
                  hx_ksk** out)
{
  hx_ksk* k = new hx_ksk{ctx, ndig, std::vector<int>(row_idx, row_idx + nrows), {}, {}};
  const size_t n = (size_t)ndig * (size_t)nrows * (size_t)ctx->phim;
  k->b.assign(b, b + n);
  k->a.assign(a, a + n);
  *out = k;
  return HX_OK;
}
int hx_ksk_shape(const hx_ksk* k, int* ndig, int* nrows, int* row_idx_out)
{
  *ndig = k->ndig;
  *nrows = (int)k->rows.size();
  if (row_idx_out)
    for (size_t r = 0; r < k->rows.size(); r++)
      row_idx_out[r] = k->rows[r];
  return HX_OK;
}
int hx_ksk_download(const hx_ksk* k, uint64_t* b, uint64_t* a)
{
  memcpy(b, k->b.data(), k->b.size() * 8);
  memcpy(a, k->a.data(), k->a.size() * 8);
  return HX_OK;
}
int hx_ksk_destroy(hx_ksk* k)
{
  delete k;
  return HX_OK;
}
// the first D digits of W restricted to the rows `allp` (in that order)
static int select_ksk(const hx_ksk* W, const std::vector<int>& allp, int D, std::vector<uint64_t>& kb,
                      std::vector<uint64_t>& ka)
{
  const size_t N = (size_t)W->ctx->phim, nr = W->rows.size();
  if (D > W->ndig)
    return fail(HX_ERR_INVALID, "more digits than the key-switching matrix has columns");
  kb.resize((size_t)D * allp.size() * N);
  ka.resize(kb.size());
  for (int d = 0; d < D; d++)
    for (size_t r = 0; r < allp.size(); r++) {
      const int w = find(W->rows, allp[r]);
      if (w < 0)
        return fail(HX_ERR_PRIMESET, "key-switching matrix lacks a prime of the ciphertext");
      memcpy(&kb[((size_t)d * allp.size() + r) * N], &W->b[((size_t)d * nr + (size_t)w) * N], N * 8);
      memcpy(&ka[((size_t)d * allp.size() + r) * N], &W->a[((size_t)d * nr + (size_t)w) * N], N * 8);
    }
  return HX_OK;
}
int hx_tensor(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1, hx_poly* o0, hx_poly* o1,
              hx_poly* o2)
{
  if (c0->idx != c1->idx || c0->idx != d0->idx || c0->idx != d1->idx)
    return fail(HX_ERR_PRIMESET, "tensorProduct: operands on different prime sets");
  for (hx_poly* o : {o0, o1, o2}) {
    o->batch = c0->batch;
    o->reshape(c0->idx);
  }
  for (int b = 0; b < c0->batch; b++) {
    std::vector<uint64_t> a0 = c0->elem(b), a1 = c1->elem(b), b0 = d0->elem(b), b1 = d1->elem(b);
    std::vector<uint64_t> r0(a0.size()), r1(a0.size()), r2(a0.size());
    ho_tensor(c0->ctx->o, c0->idx.data(), c0->nrows(), a0.data(), a1.data(), b0.data(), b1.data(), r0.data(),
              r1.data(), r2.data());
    o0->put(b, r0);
    o1->put(b, r1);
    o2->put(b, r2);
  }
  return HX_OK;
}
int hx_tensor_bring_to_set_norms(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1, hx_poly* o0,
                                 hx_poly* o1, hx_poly* o2, const int* add_idx, int nadd, const int* drop_idx, int ndrop,
                                 uint64_t ptxt_space, double* norms)
{
  int rc = hx_tensor(c0, c1, d0, d1, o0, o1, o2);
  if (rc)
    return rc;
  hx_poly* os[3] = {o0, o1, o2};
  return hx_bring_to_set_multi_norms(os, 3, add_idx, nadd, drop_idx, ndrop, ptxt_space, norms);
}
int hx_tensor_bring_to_set(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1, hx_poly* o0,
                           hx_poly* o1, hx_poly* o2, const int* add_idx, int nadd, const int* drop_idx, int ndrop,
                           uint64_t ptxt_space)
{
  return hx_tensor_bring_to_set_norms(c0, c1, d0, d1, o0, o1, o2, add_idx, nadd, drop_idx, ndrop, ptxt_space, nullptr);
}
int hx_key_switch_digits(const hx_poly* digits, const hx_ksk* W, hx_poly* out0, hx_poly* out1)
{
  const std::vector<int>& allp = out0->idx;
  if (allp != out1->idx || allp.empty() || digits->nrows() % (int)allp.size())
    return fail(HX_ERR_PRIMESET, "keySwitchDigits: digit block does not match the output parts");
  const int D = digits->nrows() / (int)allp.size();
  std::vector<uint64_t> kb, ka;
  int rc = select_ksk(W, allp, D, kb, ka);
  if (rc)
    return rc;
  for (int b = 0; b < out0->batch; b++) {
    std::vector<uint64_t> dg = digits->elem(b), p0 = out0->elem(b), p1 = out1->elem(b);
    ho_key_switch_digits(W->ctx->o, allp.data(), (int)allp.size(), D, dg.data(), kb.data(), ka.data(), p0.data(),
                         p1.data());
    out0->put(b, p0);
    out1->put(b, p1);
  }
  return HX_OK;
}
int hx_relinearize_norms(const hx_poly* t0, const hx_poly* t1, const hx_poly* t2, const hx_ksk* W, const int* dig_idx,
                         const int* dig_off, int ndig, const int* sp_idx, int nsp, hx_poly* out0, hx_poly* out1,
                         double* norms)
{
  hx_poly s0 = *t0, s1 = *t0, dg{t0->ctx, t0->batch, {}, {}};
  if (t1)
    s1 = *t1;
  else
    std::fill(s1.d.begin(), s1.d.end(), 0);
  int rc = hx_add_primes_and_scale(&s0, sp_idx, nsp);
  if (!rc)
    rc = hx_add_primes_and_scale(&s1, sp_idx, nsp);
  if (!rc)
    rc = hx_break_into_digits_norms(t2, dig_idx, dig_off, ndig, sp_idx, nsp, &dg, norms);
  if (!rc)
    rc = hx_key_switch_digits(&dg, W, &s0, &s1);
  if (rc)
    return rc;
  out0->batch = out1->batch = t0->batch;
  out0->idx = s0.idx;
  out0->d = s0.d;
  out1->idx = s1.idx;
  out1->d = s1.d;
  return HX_OK;
}
int hx_relinearize(const hx_poly* t0, const hx_poly* t1, const hx_poly* t2, const hx_ksk* W, const int* dig_idx,
                   const int* dig_off, int ndig, const int* sp_idx, int nsp, hx_poly* out0, hx_poly* out1)
{
  return hx_relinearize_norms(t0, t1, t2, W, dig_idx, dig_off, ndig, sp_idx, nsp, out0, out1, nullptr);
}
int hx_mul_relin(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1, const hx_ksk* W,
                 const int* dig_idx, const int* dig_off, int ndig, hx_poly* out0, hx_poly* out1)
{
  hx_poly t0{c0->ctx, c0->batch, {}, {}}, t1 = t0, t2 = t0;
  int rc = hx_tensor(c0, c1, d0, d1, &t0, &t1, &t2);
  if (rc)
    return rc;
  std::vector<int> sp;
  for (int i : W->rows)
    if (find(c0->idx, i) < 0)
      sp.push_back(i);
  return hx_relinearize(&t0, &t1, &t2, W, dig_idx, dig_off, ndig, sp.data(), (int)sp.size(), out0, out1);
}
int hx_mul_relin_norms(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1, const hx_ksk* W,
                       const int* dig_idx, const int* dig_off, int ndig, hx_poly* out0, hx_poly* out1, double* norms)
{
  hx_poly t0{c0->ctx, c0->batch, {}, {}}, t1 = t0, t2 = t0;
  int rc = hx_tensor(c0, c1, d0, d1, &t0, &t1, &t2);
  if (rc)
    return rc;
  std::vector<int> sp;
  for (int i : W->rows)
    if (find(c0->idx, i) < 0)
      sp.push_back(i);
  return hx_relinearize_norms(&t0, &t1, &t2, W, dig_idx, dig_off, ndig, sp.data(), (int)sp.size(), out0, out1, norms);
}

int hx_norms_flush(hx_ctx* c)
{
  for (auto& pv : c->pending)
    std::copy(pv.second.begin(), pv.second.end(), pv.first);
  c->pending.clear();
  return HX_OK;
}
int hx_ctx_defer_norms(hx_ctx* c, int on)
{
  c->defer = on != 0;
  return on ? HX_OK : hx_norms_flush(c);
}
int hx_embedding_norm(hx_ctx* ctx, const double* f_host, int rows, double* norms_out)
{
  for (int r = 0; r < rows; r++)
    norms_out[r] = ho_embedding_largest_coeff(ctx->m, f_host + (size_t)r * (size_t)ctx->phim, ctx->phim);
  return hx_norms_flush(ctx);   // host in, host out: always complete on return (and so is everything queued before)
}

int hx_ctx_timer_begin(hx_ctx*) { return HX_OK; }
int hx_ctx_timer_end(hx_ctx*, float* ms)
{
  *ms = 0;
  return HX_OK;
}
int hx_ctx_graph_begin(hx_ctx*) { return fail(HX_ERR_UNSUPPORTED, "no graphs in the CPU mock"); }
int hx_ctx_graph_end(hx_ctx*, hx_graph**) { return fail(HX_ERR_UNSUPPORTED, "no graphs in the CPU mock"); }
int hx_graph_launch(hx_graph*) { return fail(HX_ERR_UNSUPPORTED, "no graphs in the CPU mock"); }
int hx_graph_destroy(hx_graph*) { return HX_OK; }

// the HEXL-shim layer (src/intelExt.h:20-59) over the oracle's restatement of HEXL's reference transform:
// bit-reversed evaluation order and MinimalPrimitiveRoot(2n, q), what the reference's call sites assume
// (src/CModulus.cpp:385 + :421-426, :510-514) and what the engine's hx_intel_* documents
static int shim_ntt(long* out, const long* in, long n, long q, bool inverse)
{
  if (n < 2 || (n & (n - 1)) || q < 3)
    return fail(HX_ERR_INVALID, "intel shim: n must be a power of two");
  if (!ho_hexl_minimal_primitive_root((uint64_t)q, 2 * (uint64_t)n))
    return fail(HX_ERR_INVALID, "intel shim: no 2n-th root of unity modulo q");
  std::vector<uint64_t> a((size_t)n), b((size_t)n);
  for (long i = 0; i < n; i++)
    a[(size_t)i] = (uint64_t)in[i];
  if (inverse)
    ho_hexl_inverse(b.data(), a.data(), n, (uint64_t)q);
  else
    ho_hexl_forward(b.data(), a.data(), n, (uint64_t)q);
  for (long i = 0; i < n; i++)
    out[i] = (long)b[(size_t)i];
  return HX_OK;
}
int hx_intel_FFTFwd(long* out, const long* in, long n, long q) { return shim_ntt(out, in, n, q, false); }
int hx_intel_FFTRev1(long* out, const long* in, long n, long q) { return shim_ntt(out, in, n, q, true); }
#define SHIM_BIN(NAME, FN)                                                                   \
  int NAME(long* r, const long* a, const long* b, long n, long q)                            \
  {                                                                                          \
    FN((uint64_t*)r, (const uint64_t*)a, (const uint64_t*)b, n, (uint64_t)q);                \
    return HX_OK;                                                                            \
  }
#define SHIM_SC(NAME, FN)                                                                    \
  int NAME(long* r, const long* a, long s, long n, long q)                                   \
  {                                                                                          \
    long t = s % q;                                                                          \
    FN((uint64_t*)r, (const uint64_t*)a, (uint64_t)(t < 0 ? t + q : t), n, (uint64_t)q);     \
    return HX_OK;                                                                            \
  }
SHIM_BIN(hx_intel_EltwiseAddMod, ho_row_add)
SHIM_BIN(hx_intel_EltwiseSubMod, ho_row_sub)
SHIM_BIN(hx_intel_EltwiseMultMod, ho_row_mul)
SHIM_SC(hx_intel_EltwiseAddModScalar, ho_row_add_scalar)
SHIM_SC(hx_intel_EltwiseSubModScalar, ho_row_sub_scalar)
SHIM_SC(hx_intel_EltwiseMultModScalar, ho_row_mul_scalar)

}  // extern "C"

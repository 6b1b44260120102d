// engine.hip -- host runtime + C ABI (include/helib_amd.h) of the MI355X-native
// DoubleCRT engine.  Device memory, tables and launches only: there is no CPU
// compute path -- if no gfx950 device is usable every entry point fails.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/helib_amd.h"
#include "arena.h"
#include "prof.h"
#include "switches.h"
#include "dev_common.h"
#include "hostmath.h"
#include "bluestein.h"
#include "pfa_dev.h"
#include "rns_kernels.h"
#include "mfma_ext.h"
#include "rns_mfma_dev.h"
#include "norm_kernels.h"
#include "prg_kernels.h"

namespace hx {
hipError_t launch_ntt_pow2(int logn, bool inverse, const uint64_t* in, uint64_t* out,
                           const NttRows& rows, int nrows, int batch, const PrimeDev* primes,
                           const TW* tw_arena, hipStream_t st);
hipError_t launch_ntt_pow2_lazy_in(int logn, const uint64_t* in, uint64_t* out, const NttRows& rows, int nrows, int batch,
                                   const PrimeDev* primes, const TW* tw_arena, hipStream_t st);
hipError_t launch_ntt_half15_fwd(bool lazy_in, const uint64_t* in, uint64_t* out, const NttRows& rows, int nrows, int batch,
                                 const PrimeDev* sub_primes, const TW* tw_arena, hipStream_t st);
hipError_t launch_moddown_pow2(int logn, const PolyBases& data, const PolyBases& out, int drop_row,
                               int drop_prime, const NttRows& keep, int nkeep, int batch,
                               const ModDownPrep& P, const ModDownApply& A, const PrimeDev* primes,
                               const TW* tw_arena, hipStream_t st);
hipError_t launch_moddown_prep_pow2(int logn, const PolyBases& polys, int drop_row, int drop_prime, int batch,
                                    const ModDownPrep& P, const PrimeDev* primes, const TW* tw_arena,
                                    hipStream_t st);
hipError_t launch_moddown_prep_multi_pow2(int logn, const PolyBases& polys, const PrepMulti& M, int ndrop, int batch,
                                          const ModDownPrep& P, const PrimeDev* primes, const TW* tw_arena,
                                          hipStream_t st);
hipError_t launch_moddown_apply_plain_pow2(int logn, const PolyBases& polys, const PolyBases& outs,
                                           const NttRows& keep, int nkeep, int batch, const ModDownApply& A,
                                           const PrimeDev* primes, const TW* tw_arena, hipStream_t st);
hipError_t launch_moddown_prep_multi_tensor_pow2(int logn, const TensorSrc& T, const PrepMulti& M, int ndrop, int batch,
                                                 const ModDownPrep& P, const PrimeDev* primes, const TW* tw_arena,
                                                 hipStream_t st);
hipError_t launch_moddown_apply_plain_tensor_pow2(int logn, const TensorSrc& T, const PolyBases& outs, const NttRows& keep,
                                                  int nkeep, int batch, const ModDownApply& A, const PrimeDev* primes,
                                                  const TW* tw_arena, hipStream_t st);
hipError_t launch_moddown_tensor_pow2(int logn, const TensorSrc& T, const PolyBases& outs, int drop_row, int drop_prime,
                                      const NttRows& keep, int nkeep, int batch, const ModDownPrep& P,
                                      const ModDownApply& A, const PrimeDev* primes, const TW* tw_arena, hipStream_t st);
hipError_t launch_ntt_inv_mul_pow2(int logn, const uint64_t* a, const uint64_t* b, uint64_t* out, const NttRows& rows,
                                   int nrows, int batch, const PrimeDev* primes, const TW* tw_arena, hipStream_t st);
hipError_t launch_conv_rows(int logn, const ConvRowArgs& A, const ConvRows& R, int nunits, const PrimeDev* cprimes,
                            const TW* tw_arena, hipStream_t st);
}

using hx::ExtArgs;
using hx::ExtPlanDev;
using hx::MAX_ROWS;
static constexpr int MAX_EXT_SRC = hx::EXT_MAXSRC;  // source primes of one exact basis extension (generic kernel)
using hx::ModDownApply;
using hx::ModDownPrep;
using hx::NttRows;
using hx::PolyBases;
using hx::PrimeDev;
using hx::RowMap;
using hx::RowMap2;
using hx::RowScalars;
using hx::TW;

// ------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      (void)hipGetLastError(); /* reported here: do not leave it for an unrelated check */ \
      return fail(HX_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                  __FILE__, __LINE__);                                                   \
    }                                                                                    \
  } while (0)
#define CHK(expr)          \
  do {                     \
    int _rc = (expr);      \
    if (_rc != HX_OK)      \
      return _rc;          \
  } while (0)

extern "C" const char* hx_last_error(void) { return g_err.c_str(); }
extern "C" const char* hx_version(void) { return "helib_amd 0.1 (gfx950)"; }
extern "C" int hx_device_count(int* count)
{
  if (!count)
    return fail(HX_ERR_INVALID, "null argument");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(HX_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return HX_OK;
}

// ------------------------------------------------------------------
// objects
// ------------------------------------------------------------------
struct PrimeHost {
  uint64_t q, root, rinv;
  uint64_t last_s = 0, last_n = 0;  // inverse table slots 0 / 31: S0*N^-1 and N^-1
  uint64_t tw_fwd_off = 0, tw_inv_off = 0;  // into hx_ctx::d_tw (TW units)
  bool proth = false;  // row tables in Proth form (8-byte entries w 2^64 mod q): PrimeDev::proth
  int half_pd = -1;    // N = 2^15: index into hx_ctx::d_cprimes of the forward half-row kernel's entry for g = 0 (g = 1 follows)
};
// the second word of a twiddle constant that a row kernel substitutes for a table entry (ModDownPrep::upS / upN):
// Shoup's quotient, or -- for a prime whose rows run the Proth-form butterflies -- w 2^64 mod q
static inline uint64_t tw_companion(const PrimeHost& ph, uint64_t w)
{
  return ph.proth ? hx::tw_mont_form(w, ph.q) : hxh::shoup(w, ph.q);
}
// w 2^128 mod q for a Proth-form prime (the constant that multiplies a value reduced by mont_redc128), else 0
static inline uint64_t tw_r2(const PrimeHost& ph, uint64_t w)
{
  return ph.proth ? hx::tw_mont_form(hx::tw_mont_form(w, ph.q), ph.q) : 0;
}

struct ExtPlan {
  ExtPlanDev dev;
  void* blob = nullptr;
  void* blob_mfma = nullptr;   // the matrix-core form's tables (mfma_ext.h), when the plan has them
};

struct ConvPlan {  // negacyclic NTT of size 2^logn for one prime
  int logn = 0;
  int split = 0;             // 0: one row transform; 4 / 8 / 16: radix-4 / -8 / -16 split into sub-transforms
  int pd[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // indices into hx_ctx::d_cprimes
  hx::ConvPrimeDev* dev = nullptr;
};
struct BluePrime {
  // aux: q-1 lacks the 2-power roots the chirp convolutions need; they run modulo three auxiliary
  // NTT primes and are recombined exactly (bluestein.h, crt3_kernel)
  bool aux = false;
  ConvPlan aconv[3][3];                                       // [aux prime][size]
  uint64_t* d_ahat[3][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr},
                            {nullptr, nullptr, nullptr, nullptr}};
  hx::Crt3Dev crt;
  hx::BluePrimeDev* dev = nullptr;
  TW* d_powers = nullptr;
  TW* d_ipowers = nullptr;
  uint64_t* d_hat[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // Rb, iRb, NTT(-Psi), NTT(Phi), NTT(Phi mod X^(2^n3) + 1)
  TW* d_hatw[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};       // the same as Shoup pairs (fused convolution kernel)
  ConvPlan conv[4];                                           // sizes bk, n1, n2, n3 (n3: the aliased form of the last one)
  uint64_t* d_pfa = nullptr;   // m = 21845 on a Proth-form prime: the Good-Thomas x Rader tables (pfa_core.h), else null
};

struct hx_ctx {
  hxs::Switches sw;  // the environment switches as they were when the context was created (switches.h)
  int device = 0;
  hipStream_t stream = nullptr;
  uint64_t m = 0;
  uint32_t phim = 0;
  int logn = 0;  // log2(phim) when m is a power of two, else 0
  bool pow2 = false;
  std::vector<uint32_t> zms;
  uint32_t* d_zms = nullptr;
  int32_t* d_zms_index = nullptr;
  uint32_t* d_perm = nullptr;
  std::vector<PrimeHost> primes;
  PrimeDev* d_primes = nullptr;
  int primes_cap = 0;
  TW* d_tw = nullptr;  // twiddle arena shared by all primes
  size_t tw_cap = 0, tw_used = 0;
  uint64_t* scratch[11] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t* d_redo = nullptr;   // redo list of the HPS-form RNS kernels: [0] = count, [1..] = coefficient indices
  size_t redo_cap = 0;          // (in 32-bit words)
  size_t scratch_words[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint64_t aux_q[3] = {0, 0, 0};  // auxiliary NTT primes of the aux Bluestein path (lazily chosen)
  // general m (Bluestein): conv sizes 2^bk (chirp), 2^n1 / 2^n2 (rem Phi_m), pseudo "primes"
  // (twiddle tables of the conv sizes) and per-prime tables
  int bk = 0, n1 = 0, n2 = 0, n3 = 0;   // n3 < n2: rem Phi_m's product Q Phi_m taken modulo X^(2^n3) + 1 (bluestein_rows)
  uint32_t dq = 0, mpad = 0;  // dq = m-1-phi(m) = deg of the quotient by Phi_m
  PrimeDev* d_cprimes = nullptr;
  int ncprimes = 0, cprimes_cap = 0;
  std::vector<struct BluePrime*> blue;
  uint16_t* d_pfa_idx = nullptr;   // m = 21845: pos2 [16384], dlog3 [257 (+1 pad)], gpow3 [256]  (pfa::host::build_index_tables)
  std::vector<ConvPlan> bigplan;  // power-of-two rings with N = 2^16..2^18: per-prime split plans (pow2_big_rows)
  std::vector<int64_t> psi_low, phi_coef;  // -Psi mod X^(dq+1) and Phi_m, small integers
  std::map<std::vector<uint64_t>, ExtPlan*> plans;
  // stream-ordered device-memory arena for poly slabs (arena.h): a few large hipMalloc chunks,
  // sub-allocated; every slab is used on the context's one stream only, so a released extent can be
  // handed out again without synchronising (hipMalloc / hipFree would serialise the device on every
  // DoubleCRT temporary, and results that are kept alive would reach hipMalloc on every operation).
  hxa::SlabArena arena;
  std::unordered_set<struct hx_poly*> polys;  // every live poly of this context (graph capture un-shares them)
  // lifetime: polys and key-switch matrices keep their context alive, so the
  // handles may be destroyed in any order (hx_ctx_destroy only drops the
  // caller's reference).
  int refs = 1;
  std::recursive_mutex mu;   // see CTX_ENTER
  // canonical-embedding norms (N1): when frac is non-null the exact-RNS kernels also emit
  // value/P as doubles there (one [batch][N] block per poly / digit, starting at frac_pos)
  double* d_frac = nullptr;
  size_t frac_cap = 0, frac_pos = 0;
  bool want_frac = false;
  // the fused single-prime mod-down leaves (x, S) in scratch[0]/[1]: the norm kernel reads them
  // directly (rows [xs_first, xs_first + xs_rows) of the norm batch, delta/qd = x/qd - S)
  int xs_first = 0, xs_rows = 0;
  double xs_inv_qd = 0;
  bool want_fdelta = false;  // the caller also wants the fdelta coefficients on the host
  double2* d_wtab = nullptr;           // W^k, k < N, W = exp(2 pi i / m)  (m a power of two)
  // general m (complex-double Bluestein, norm_kernels.h): v_k, exp(2 pi i k/P) (k < P/2), transform
  // of the chirp, and the transform-domain work buffer
  double2 *d_bn_v = nullptr, *d_bn_w = nullptr, *d_bn_chat = nullptr, *d_bn_Z = nullptr;
  size_t bn_rows_cap = 0;
  unsigned long long* d_norm2 = nullptr;
  size_t norm_cap = 0;
  double2* d_norm_park = nullptr;      // embed_norm_quarter_split_kernel: the parked sub-transforms
  size_t norm_park_cap = 0;
  // deferred read-back of norms (hx_ctx_defer_norms): squared norms land in pinned host slots,
  // an event marks each; hx_norms_flush converts them into the callers' arrays
  struct NormPending {
    hipEvent_t ev;
    unsigned long long* pinned;
    size_t cap;
    int rows;
    double* out;
    bool slab;   // pinned points into one of norm_slabs (freed with it)
  };
  // pinned read-back buffers come 64 at a time (one hipHostMalloc per 64 read-backs kept pending, not one each)
  std::vector<void*> norm_slabs;
  bool defer_norms = false;
  std::vector<NormPending> norm_pending, norm_free;
  // The norm kernels run on a stream of their own, next to whatever the context enqueues after them
  // (they have no consumer on the device: only the host reads the result).  One LDS-bound workgroup per
  // CU with almost no VALU work co-resides with the HBM-bound tensor / key-switch kernels.  Ordering:
  // the side stream waits for an event recorded behind the producers of its input; the main stream
  // waits for the side stream (norm_join) before anything rewrites what an in-flight norm kernel reads:
  // scratch[0] / [1] (the mod-down's x, S) or d_frac.
  hipStream_t norm_stream = nullptr;
  hipEvent_t norm_in_ev = nullptr, norm_out_ev = nullptr;
  bool norm_reads_xs = false, norm_reads_frac = false;
  hipEvent_t timer[2] = {nullptr, nullptr};  // hx_ctx_timer_begin / _end
  // HIP graphs (hx_ctx_graph_begin / _end): while a capture is open or a captured graph is alive,
  // nothing a graph may point at is handed back -- slabs that were live during a capture are pinned
  // in the arena and wait in its deferred list when their poly lets go of them; buffers that are
  // re-grown (scratch, twiddle arena, tables) are retired instead of freed
  bool capturing = false;
  int graphs_alive = 0;
  hipStream_t own_stream = nullptr;   // created by the first capture of a context that ran on the default stream
  std::vector<void*> graph_retired;
};

struct hx_poly {
  hx_ctx* ctx;
  int batch;
  int cap_rows;
  std::vector<int> prime_idx;  // one per row
  uint64_t* d;
  bool owns;
  // Copy-on-write: hx_poly_copy between pool-backed polys shares the source's slab (all sharers
  // have the same d / cap_rows) and bumps this count; whoever writes first takes a private copy
  // (poly_own) -- except the fused mod-switch, which reads the shared slab and writes its result
  // straight into a fresh one, so that `Ctxt tmp = other; tmp.bringToSet(...)`
  // (src/Ctxt.cpp:1700-1745) never moves the operand twice.
  struct Share {
    int refs;
  };
  Share* share = nullptr;
  // the caller holds a raw device pointer into this poly's slab (hx_poly_device_ptr): copies from and
  // to it are made eagerly from then on, so that writes through the pointer never reach a "copy"
  bool exposed = false;
  size_t row_words() const { return (size_t)batch * ctx->phim; }
  int nrows() const { return (int)prime_idx.size(); }
  bool shared() const { return share && share->refs > 1; }
};

struct hx_ksk {
  hx_ctx* ctx;
  int ndig;
  std::vector<int> row_idx;
  uint64_t* d_b;
  uint64_t* d_a;
};

static int use(hx_ctx* c)
{
  HIPCHK(hipSetDevice(c->device));
  return HX_OK;
}
// Every entry point holds its context's (recursive) mutex while it touches host-side state -- the
// plan cache, the slab pool, scratch slots, prime tables -- and enqueues its kernels; the device work
// itself is ordered by the context's single stream.  This makes calls on distinct polys of ONE
// context safe from several threads, which is how HElib's NTL thread pool uses DoubleCRT objects
// (re-entrant on distinct objects, src/CModulus.cpp:580-610).
#define CTX_ENTER(ctx)                                             \
  std::lock_guard<std::recursive_mutex> _ctx_lock((ctx)->mu);      \
  CHK(use(ctx))

// entry points that must wait for the device cannot be recorded into a graph: they fail before
// touching HIP, so that the open capture stays valid
#define NO_CAPTURE(ctx, what)                                                                      \
  do {                                                                                             \
    if ((ctx)->capturing)                                                                          \
      return fail(HX_ERR_INVALID, what " waits for the device and cannot be captured in a graph"); \
  } while (0)

static constexpr size_t POOL_LIMIT = (size_t)64 << 30;        // keep at most 64 GiB cached
template <bool TRACE>   // (HX_ARENA_TRACE) one line per hipMalloc the arena makes
static int arena_sys_alloc_t(size_t bytes, void** out)
{
  const bool trace = TRACE;
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(out, bytes);
  if (trace)
    fprintf(stderr, "[helib_amd arena] hipMalloc(%zu MiB) took %.2f ms\n", bytes >> 20,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  if (e != hipSuccess)
    (void)hipGetLastError();
  return (int)e;
}
static void arena_sys_free(void* p) { hipFree(p); }  // (hipFree waits for the device)
static hipError_t pool_alloc(hx_ctx* c, size_t bytes, void** out)
{
  if (!c->arena.sys_alloc) {
    c->arena.sys_alloc = c->sw.arena_trace ? arena_sys_alloc_t<true> : arena_sys_alloc_t<false>;
    c->arena.sys_free = arena_sys_free;
  }
  // a block taken while a capture is open may end up inside the graph: pinned from the start
  int rc = c->arena.alloc(bytes, c->capturing, out);
  if (rc != 0 && !c->capturing) {
    // out of device memory: give the arena's empty chunks back and try once more -- explicitly after the stream has
    // drained (extents of those chunks may have been released behind work still in flight), never inside a capture
    (void)hipGetLastError();
    hipStreamSynchronize(c->stream);
    if (c->arena.trim(0) > 0)
      rc = c->arena.alloc(bytes, false, out);
  }
  return (hipError_t)rc;
}
// a device buffer that is being replaced by a larger one: freed now, or kept for the graphs that
// may have its address baked in (released when the last of them is destroyed)
static void retire_or_free(hx_ctx* c, void* p, bool device_wide_sync = false)
{
  if (!p)
    return;
  if (c->norm_stream)
    hipStreamSynchronize(c->norm_stream);  // (the buffer may be one the norm kernels use)
  if (c->capturing || c->graphs_alive > 0) {
    c->graph_retired.push_back(p);
    return;
  }
  if (device_wide_sync)
    hipDeviceSynchronize();
  else
    hipStreamSynchronize(c->stream);
  hipFree(p);
}
static void pool_free(hx_ctx* c, void* p, size_t bytes)
{
  (void)bytes;
  // only what a graph can point at waits for the graphs to go: blocks that were live while a capture
  // was open.  Slabs of eager work next to a live graph recycle at once (no growth).
  if ((c->capturing || c->graphs_alive > 0) && c->arena.is_pinned(p)) {
    c->arena.defer(p);
    return;
  }
  c->arena.release(p);
  if (c->arena.cached() > POOL_LIMIT && !c->capturing) {
    hipStreamSynchronize(c->stream);
    c->arena.trim(POOL_LIMIT / 2);
  }
}

// the main stream waits for the norm kernels still reading the (x, S) scratch slots / d_frac
static int norm_join(hx_ctx* c, bool xs, bool frac)
{
  if ((xs && c->norm_reads_xs) || (frac && c->norm_reads_frac)) {
    HIPCHK(hipStreamWaitEvent(c->stream, c->norm_out_ev, 0));
    c->norm_reads_xs = c->norm_reads_frac = false;   // (one event: everything on the side stream so far)
  }
  return HX_OK;
}

static int ensure_scratch(hx_ctx* c, int slot, size_t words)
{
  if (slot <= 1)  // about to be rewritten by the caller
    CHK(norm_join(c, true, false));
  if (c->scratch_words[slot] >= words)
    return HX_OK;
  if (c->scratch[slot]) {
    retire_or_free(c, c->scratch[slot]);
    c->scratch[slot] = nullptr;
    c->scratch_words[slot] = 0;
  }
  HIPCHK(hipMalloc((void**)&c->scratch[slot], words * 8));
  c->scratch_words[slot] = words;
  return HX_OK;
}

// stream-ordered device-to-device copy of `words` 64-bit words by a kernel (see copy_words_kernel)
static int dcopy(hx_ctx* c, uint64_t* dst, const uint64_t* src, size_t words)
{
  if (words == 0 || dst == src)
    return HX_OK;
  if (((uintptr_t)dst | (uintptr_t)src) & 15) {  // never the case for pool slabs and whole rows
    HIPCHK(hipMemcpyAsync(dst, src, words * 8, hipMemcpyDeviceToDevice, c->stream));
    return HX_OK;
  }
  const size_t blocks = std::min<size_t>((words / 2 + 255) / 256, (size_t)256 * 16);
  HX_LAUNCH(hx::copy_words_kernel, dim3((unsigned)std::max<size_t>(blocks, 1)), dim3(256), 0, c->stream,
                     dst, src, words);
  HIPCHK(hipGetLastError());
  return HX_OK;
}

// give up p's slab (pool memory): the last holder returns it to the pool
static void storage_release(hx_poly* p)
{
  if (!p->owns || !p->d)
    return;
  const size_t bytes = (size_t)p->cap_rows * p->row_words() * 8;
  if (p->share) {
    if (--p->share->refs == 0) {
      pool_free(p->ctx, p->d, bytes);
      delete p->share;
    }
    p->share = nullptr;
  } else {
    pool_free(p->ctx, p->d, bytes);
  }
  p->d = nullptr;
}
// make p the only holder of its slab (every entry point that writes p's rows calls this first)
static int poly_own(hx_poly* p)
{
  if (!p->share)
    return HX_OK;
  if (p->share->refs == 1) {
    delete p->share;
    p->share = nullptr;
    return HX_OK;
  }
  hx_ctx* c = p->ctx;
  uint64_t* nd = nullptr;
  const size_t bytes = (size_t)p->cap_rows * p->row_words() * 8;
  hipError_t e = pool_alloc(c, bytes, (void**)&nd);
  if (e != hipSuccess)
    return fail(HX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  CHK(dcopy(c, nd, p->d, (size_t)p->nrows() * p->row_words()));
  p->share->refs--;
  p->share = nullptr;
  p->d = nd;
  return HX_OK;
}
#define OWN(p) CHK(poly_own(p))
// p is about to have every live row overwritten by kernels that can read the old rows from another
// address: a shared p moves to a fresh slab WITHOUT copying; *old_rows is where the old rows are
// (the shared slab stays with its other holders; reuse of pool memory is stream-ordered, so the
// kernels enqueued by this entry point still see it even if those holders are destroyed next).
static int poly_fresh(hx_poly* p, const uint64_t** old_rows)
{
  *old_rows = p->d;
  if (!p->share)
    return HX_OK;
  if (p->share->refs == 1) {
    delete p->share;
    p->share = nullptr;
    return HX_OK;
  }
  uint64_t* nd = nullptr;
  const size_t bytes = (size_t)p->cap_rows * p->row_words() * 8;
  hipError_t e = pool_alloc(p->ctx, bytes, (void**)&nd);
  if (e != hipSuccess)
    return fail(HX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  p->share->refs--;
  p->share = nullptr;
  p->d = nd;
  return HX_OK;
}

// ------------------------------------------------------------------
// context
// ------------------------------------------------------------------
static void ctx_free(hx_ctx* c);
extern "C" int hx_ctx_create(hx_ctx** out, int device, uint64_t m)
{
  if (!out || m < 2 || m > (1ull << 24))
    return fail(HX_ERR_INVALID, "Bad Z_m^* modulus m (must be in [2, 2^24])");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(HX_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
  if (device < 0 || device >= ndev)
    return fail(HX_ERR_INVALID, "device %d out of range [0,%d)", device, ndev);
  HIPCHK(hipSetDevice(device));
  hx_ctx* c = new hx_ctx();
  c->sw = hxs::read();   // the environment switches (switches.h), snapshotted here and nowhere else
  // a failing allocation below returns through HIPCHK: the half-built context and whatever it
  // already holds on the device go with it
  struct Guard {
    hx_ctx* c;
    ~Guard()
    {
      if (c)
        ctx_free(c);
    }
  } guard{c};
  c->device = device;
  c->m = m;
  std::vector<int32_t> zidx((size_t)m, -1);
  for (uint64_t i = 1; i < m; i++)
    if (hxh::gcd(i, m) == 1) {
      zidx[i] = (int32_t)c->zms.size();
      c->zms.push_back((uint32_t)i);
    }
  c->phim = (uint32_t)c->zms.size();
  c->pow2 = (m & (m - 1)) == 0;
  if (c->pow2) {
    int l = 0;
    while ((1u << l) < c->phim)
      l++;
    c->logn = l;
  }
  HIPCHK(hipMalloc((void**)&c->d_zms, (size_t)c->phim * 4));
  HIPCHK(hipMalloc((void**)&c->d_zms_index, (size_t)m * 4));
  HIPCHK(hipMalloc((void**)&c->d_perm, (size_t)c->phim * 4));
  HIPCHK(hipMemcpy(c->d_zms, c->zms.data(), (size_t)c->phim * 4, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_zms_index, zidx.data(), (size_t)m * 4, hipMemcpyHostToDevice));
  c->primes_cap = 1024;
  HIPCHK(hipMalloc((void**)&c->d_primes, sizeof(PrimeDev) * c->primes_cap));
  guard.c = nullptr;
  *out = c;
  return HX_OK;
}

static void ctx_release(hx_ctx* c)
{
  bool last;
  {
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    last = --c->refs == 0;
  }
  if (last)  // nobody else can hold a handle into this context any more
    ctx_free(c);
}
extern "C" int hx_ctx_destroy(hx_ctx* c)
{
  if (!c)
    return HX_OK;
  ctx_release(c);
  return HX_OK;
}
static void ctx_free(hx_ctx* c)
{
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  if (c->d_tw)
    hipFree(c->d_tw);
  for (auto& kv : c->plans) {
    hipFree(kv.second->blob);
    hipFree(kv.second->blob_mfma);
    delete kv.second;
  }
  for (int i = 0; i < 11; i++)
    if (c->scratch[i])
      hipFree(c->scratch[i]);
  c->arena.destroy();
  if (c->norm_stream) {
    hipStreamDestroy(c->norm_stream);
    hipEventDestroy(c->norm_in_ev);
    hipEventDestroy(c->norm_out_ev);
  }
  if (c->own_stream)
    hipStreamDestroy(c->own_stream);
  for (void* q : c->graph_retired)
    hipFree(q);
  hipFree(c->d_redo);
  hipFree(c->d_frac);
  hipFree(c->d_wtab);
  hipFree(c->d_bn_v);
  hipFree(c->d_bn_w);
  hipFree(c->d_bn_chat);
  hipFree(c->d_bn_Z);
  hipFree(c->d_norm2);
  hipFree(c->d_norm_park);
  for (auto* v : {&c->norm_pending, &c->norm_free})
    for (auto& np : *v) {
      hipEventDestroy(np.ev);
      if (!np.slab)
        hipHostFree(np.pinned);
    }
  for (void* q : c->norm_slabs)
    hipHostFree(q);
  for (BluePrime* b : c->blue) {
    if (!b)
      continue;
    hipFree(b->dev);
    hipFree(b->d_powers);
    hipFree(b->d_ipowers);
    hipFree(b->d_pfa);
    for (int i = 0; i < 5; i++) {
      hipFree(b->d_hat[i]);
      hipFree(b->d_hatw[i]);
    }
    for (int i = 0; i < 4; i++)
      hipFree(b->conv[i].dev);
    for (int j = 0; j < 3; j++) {
      for (int i = 0; i < 4; i++)
        hipFree(b->d_ahat[j][i]);
      for (int i = 0; i < 3; i++)
        hipFree(b->aconv[j][i].dev);
    }
    delete b;
  }
  if (c->d_cprimes)
    hipFree(c->d_cprimes);
  for (hipEvent_t e : c->timer)
    if (e)
      hipEventDestroy(e);
  hipFree(c->d_pfa_idx);
  hipFree(c->d_zms);
  hipFree(c->d_zms_index);
  hipFree(c->d_perm);
  hipFree(c->d_primes);
  delete c;
}

extern "C" int hx_ctx_phim(const hx_ctx* c, uint64_t* phim)
{
  if (!c || !phim)
    return fail(HX_ERR_INVALID, "null argument");
  *phim = c->phim;
  return HX_OK;
}
extern "C" int hx_ctx_set_stream(hx_ctx* c, void* s)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null context");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  if (c->capturing)
    return fail(HX_ERR_INVALID, "hx_ctx_set_stream while a graph is being captured");
  if (c->stream != (hipStream_t)s) {
    // pooled slabs are recycled in stream order: drain the old stream before switching
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->norm_stream)
      hipStreamSynchronize(c->norm_stream);
  }
  c->stream = (hipStream_t)s;
  return HX_OK;
}
extern "C" int hx_ctx_sync(hx_ctx* c)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null context");
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_ctx_sync");
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->norm_stream)
    HIPCHK(hipStreamSynchronize(c->norm_stream));
  return HX_OK;
}
extern "C" int hx_ctx_reserve(hx_ctx* c, uint64_t bytes)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null context");
  CTX_ENTER(c);
  if (!c->arena.sys_alloc) {
    c->arena.sys_alloc = c->sw.arena_trace ? arena_sys_alloc_t<true> : arena_sys_alloc_t<false>;
    c->arena.sys_free = arena_sys_free;
  }
  const int rc = c->arena.reserve((size_t)bytes);
  if (rc != 0)
    return fail(HX_ERR_NOMEM, "hipMalloc(%llu) failed: %s", (unsigned long long)bytes, hipGetErrorString((hipError_t)rc));
  return HX_OK;
}
extern "C" int hx_ctx_arena_stats(hx_ctx* c, uint64_t out[4])
{
  if (!c || !out)
    return fail(HX_ERR_INVALID, "null argument");
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  out[0] = c->arena.reserved;
  out[1] = c->arena.in_use;
  out[2] = c->arena.sys_calls;
  out[3] = c->arena.deferred.size();
  return HX_OK;
}
extern "C" int hx_ctx_num_primes(const hx_ctx* c, int* n)
{
  if (!c || !n)
    return fail(HX_ERR_INVALID, "null argument");
  *n = (int)c->primes.size();
  return HX_OK;
}
extern "C" int hx_ctx_prime(const hx_ctx* c, int idx, uint64_t* q, uint64_t* root)
{
  if (!c)
    return fail(HX_ERR_INVALID, "prime index out of range");
  std::lock_guard<std::recursive_mutex> lk(const_cast<hx_ctx*>(c)->mu);
  if (idx < 0 || idx >= (int)c->primes.size())
    return fail(HX_ERR_INVALID, "prime index out of range");
  if (q)
    *q = c->primes[idx].q;
  if (root)
    *root = c->primes[idx].root;
  return HX_OK;
}

static int tw_reserve(hx_ctx* c, size_t extra)
{
  if (c->tw_used + extra <= c->tw_cap)
    return HX_OK;
  size_t ncap = c->tw_cap ? c->tw_cap * 2 : extra * 32;
  while (ncap < c->tw_used + extra)
    ncap *= 2;
  TW* nd = nullptr;
  HIPCHK(hipMalloc((void**)&nd, ncap * sizeof(TW)));
  if (c->d_tw) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(nd, c->d_tw, c->tw_used * sizeof(TW), hipMemcpyDeviceToDevice));
    retire_or_free(c, c->d_tw, true);   // (a live graph keeps using the old arena: its content stays valid)
  }
  c->d_tw = nd;
  c->tw_cap = ncap;
  return HX_OK;
}

template <int LOGN>
static int upload_tw(hx_ctx* c, PrimeHost& ph)
{
  using G = hx::Geo<LOGN>;
  std::vector<TW> f(G::TW_TOTAL), i(G::TW_TOTAL);
  uint64_t ninv = hxh::invmod((uint64_t)G::N % ph.q, ph.q);
  hx::build_tw_tables<LOGN>(ph.q, ph.root, ph.rinv, ninv, hxh::mulmod, f.data(), i.data());
  CHK(tw_reserve(c, 2 * (size_t)G::TW_TOTAL));
  ph.tw_fwd_off = c->tw_used;
  ph.tw_inv_off = c->tw_used + G::TW_TOTAL;
  // q = 1 (mod 2^32) -- every prime PrimeGenerator makes for these rings down to ~45 bits: the rows run the
  // Proth-form butterflies (ntt_core.h, ArProth) on 8-byte entries w 2^64 mod q at the same table positions
  // (each table keeps its slot of TW_TOTAL 16-byte units and fills half of it; offsets stay in TW units)
  ph.proth = hx::is_proth32(ph.q) && !c->sw.no_proth;
  if (ph.proth) {
    std::vector<hx::TWM> fm(G::TW_TOTAL), im(G::TW_TOTAL);
    hx::tw_tables_to_mont(f.data(), G::TW_TOTAL, ph.q, fm.data());
    hx::tw_tables_to_mont(i.data(), G::TW_TOTAL, ph.q, im.data());
    HIPCHK(hipMemcpy(c->d_tw + ph.tw_fwd_off, fm.data(), sizeof(hx::TWM) * G::TW_TOTAL, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_tw + ph.tw_inv_off, im.data(), sizeof(hx::TWM) * G::TW_TOTAL, hipMemcpyHostToDevice));
  } else {
    HIPCHK(hipMemcpy(c->d_tw + ph.tw_fwd_off, f.data(), sizeof(TW) * G::TW_TOTAL,
                     hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d_tw + ph.tw_inv_off, i.data(), sizeof(TW) * G::TW_TOTAL,
                     hipMemcpyHostToDevice));
  }
  c->tw_used += 2 * (size_t)G::TW_TOTAL;
  ph.last_s = i[0].w;
  ph.last_n = i[31].w;
  return HX_OK;
}

static int upload_tw_small(hx_ctx* c, PrimeHost& ph)
{
  const size_t N = (size_t)1 << c->logn;
  std::vector<TW> f(N), i(N);
  uint64_t ninv = hxh::invmod((uint64_t)N % ph.q, ph.q);
  hx::build_tw_small(c->logn, ph.q, ph.root, ph.rinv, ninv, hxh::mulmod, f.data(), i.data());
  CHK(tw_reserve(c, 2 * N));
  ph.tw_fwd_off = c->tw_used;
  ph.tw_inv_off = c->tw_used + N;
  HIPCHK(hipMemcpy(c->d_tw + ph.tw_fwd_off, f.data(), sizeof(TW) * N, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_tw + ph.tw_inv_off, i.data(), sizeof(TW) * N, hipMemcpyHostToDevice));
  c->tw_used += 2 * N;
  return HX_OK;
}

// ------------------------------------------------------------------
// general m: Bluestein tables and convolution plans
// ------------------------------------------------------------------
static int ntt_launch(hx_ctx* c, int logn, const PrimeDev* table, const uint64_t* in, uint64_t* out,
                      const std::vector<std::pair<int, int>>& rows, int batch, bool inverse, bool lazy_in = false);
static int ensure_scratch(hx_ctx* c, int slot, size_t words);

static int next_pow2_exp(uint64_t n)  // NTL::NextPowerOfTwo: least k with 2^k >= n
{
  int k = 0;
  while (((uint64_t)1 << k) < n)
    k++;
  return k;
}

// Phi_m(X) and Psi(X) = (X^m - 1)/Phi_m(X) over Z (small coefficients), by exact power-series
// arithmetic on the products of (X^d - 1)^(mu(m/d)).
static int mobius(uint64_t n)
{
  int mu = 1;
  for (uint64_t p = 2; p * p <= n; p++)
    if (n % p == 0) {
      n /= p;
      if (n % p == 0)
        return 0;
      mu = -mu;
    }
  return n > 1 ? -mu : mu;
}
static std::vector<int64_t> cyclo_product(uint64_t m, size_t len, bool want_phi)
{
  // want_phi: prod_{d|m} (X^d-1)^{mu(m/d)} ; else Psi = prod_{d|m, d<m} Phi_d = (X^m-1)/Phi_m
  // computed as (X^m - 1) * prod (X^d-1)^{-mu(m/d)}; both truncated to `len` coefficients.
  std::vector<int64_t> a(len, 0);
  a[0] = 1;
  auto mul = [&](uint64_t d) {  // *= (X^d - 1)
    for (size_t i = len; i-- > 0;) {
      int64_t v = -a[i];
      if (i >= d)
        v += a[i - d];
      a[i] = v;
    }
  };
  auto div = [&](uint64_t d) {  // /= (X^d - 1)
    for (size_t i = 0; i < len; i++) {
      int64_t v = -a[i];
      if (i >= d)
        v += a[i - d];
      a[i] = v;
    }
  };
  for (uint64_t d = 1; d <= m; d++) {
    if (m % d)
      continue;
    int mu = mobius(m / d);
    if (!want_phi)
      mu = -mu;
    if (mu == 1)
      mul(d);
  }
  if (!want_phi)
    mul(m);
  for (uint64_t d = 1; d <= m; d++) {
    if (m % d)
      continue;
    int mu = mobius(m / d);
    if (!want_phi)
      mu = -mu;
    if (mu == -1)
      div(d);
  }
  return a;
}

static int cprime_add(hx_ctx* c, uint64_t q, uint64_t fwd_off, uint64_t inv_off, int* idx, bool proth = false)
{
  if (c->ncprimes == c->cprimes_cap) {
    int ncap = c->cprimes_cap ? c->cprimes_cap * 2 : 256;
    PrimeDev* nd = nullptr;
    HIPCHK(hipMalloc((void**)&nd, sizeof(PrimeDev) * ncap));
    if (c->d_cprimes) {
      HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipMemcpy(nd, c->d_cprimes, sizeof(PrimeDev) * c->ncprimes, hipMemcpyDeviceToDevice));
      retire_or_free(c, c->d_cprimes, true);
    }
    c->d_cprimes = nd;
    c->cprimes_cap = ncap;
  }
  PrimeDev pd;
  memset(&pd, 0, sizeof pd);
  pd.q = q;
  pd.q2 = 2 * q;
  pd.k = (uint32_t)hxh::bitlen(q);
  pd.mu = (uint64_t)((((hxh::u128)1) << (2 * pd.k)) / q);
  pd.mu64 = (uint64_t)((((hxh::u128)1) << 64) / q);
  pd.mu63 = (uint64_t)((((hxh::u128)1) << (63 + pd.k)) / q);
  pd.tw_fwd_off = fwd_off;
  pd.tw_inv_off = inv_off;
  pd.proth = proth ? 1u : 0u;
  pd.r2 = hx::tw_mont_form(hx::tw_mont_form(1 % q, q), q);
  HIPCHK(hipMemcpy(c->d_cprimes + c->ncprimes, &pd, sizeof pd, hipMemcpyHostToDevice));
  *idx = c->ncprimes++;
  return HX_OK;
}

static int tw_upload(hx_ctx* c, const std::vector<TW>& f, const std::vector<TW>& i, uint64_t* fo,
                     uint64_t* io)
{
  CHK(tw_reserve(c, f.size() + i.size()));
  *fo = c->tw_used;
  *io = c->tw_used + f.size();
  HIPCHK(hipMemcpy(c->d_tw + *fo, f.data(), sizeof(TW) * f.size(), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_tw + *io, i.data(), sizeof(TW) * i.size(), hipMemcpyHostToDevice));
  c->tw_used += f.size() + i.size();
  return HX_OK;
}

template <int LOGQ>
static int conv_tables_sub(hx_ctx* c, uint64_t q, uint64_t psi, int OUT, unsigned g, int* pd_idx)
{
  using G = hx::Geo<LOGQ>;
  std::vector<TW> f(G::TW_TOTAL), i(G::TW_TOTAL);
  uint64_t ninv = hxh::invmod((uint64_t)G::N % q, q);
  hx::build_tw_tables_sub<LOGQ>(q, psi, hxh::invmod(psi, q), ninv, hxh::mulmod, OUT, g, f.data(),
                                i.data());
  uint64_t fo, io;
  // a Proth-form prime (q = 1 mod 2^32: every PrimeGenerator prime of these rings; the m = 21845 primes are
  // c 2^36 + 1): the sub-transform's rows run the Proth-form butterflies on 8-byte entries, as the primes' own row
  // tables do (upload_tw) -- in the convolution kernel, the split power-of-two rings and the round-2 chain alike
  // (all of them go through RowNTT with the arithmetic chosen by PrimeDev::proth)
  const bool proth = hx::is_proth32(q) && !c->sw.no_proth;
  if (proth) {
    std::vector<TW> fm((G::TW_TOTAL + 1) / 2), im((G::TW_TOTAL + 1) / 2);   // the same bytes as TW_TOTAL 8-byte entries
    hx::tw_tables_to_mont(f.data(), G::TW_TOTAL, q, reinterpret_cast<hx::TWM*>(fm.data()));
    hx::tw_tables_to_mont(i.data(), G::TW_TOTAL, q, reinterpret_cast<hx::TWM*>(im.data()));
    CHK(tw_upload(c, fm, im, &fo, &io));
  } else {
    CHK(tw_upload(c, f, i, &fo, &io));
  }
  return cprime_add(c, q, fo, io, pd_idx, proth);
}

// psi: primitive 2^(logn+1)-th root of unity mod q
static int conv_plan_create(hx_ctx* c, uint64_t q, int logn, uint64_t psi, ConvPlan* pl)
{
  pl->logn = logn;
  pl->split = logn > 18 ? 16 : (logn > 17 ? 8 : (logn > 15 ? 4 : 0));
  if (logn < 1 || logn > 19)
    return fail(HX_ERR_UNSUPPORTED, "convolution size 2^%d not supported (m too large)", logn);
  hx::ConvPrimeDev h;
  memset(&h, 0, sizeof h);
  h.q = q;
  h.k = (uint32_t)hxh::bitlen(q);
  h.mu = (uint64_t)((((hxh::u128)1) << (2 * h.k)) / q);
  h.mu64 = (uint64_t)((((hxh::u128)1) << 64) / q);
  h.logn = (uint32_t)logn;
  if (logn <= 12) {
    size_t N = (size_t)1 << logn;
    std::vector<TW> f(N), i(N);
    hx::build_tw_small(logn, q, psi, hxh::invmod(psi, q), hxh::invmod((uint64_t)N % q, q),
                       hxh::mulmod, f.data(), i.data());
    uint64_t fo, io;
    CHK(tw_upload(c, f, i, &fo, &io));
    CHK(cprime_add(c, q, fo, io, &pl->pd[0]));
  } else if (!pl->split) {
    switch (logn) {
      case 13: CHK(conv_tables_sub<13>(c, q, psi, 0, 0, &pl->pd[0])); break;
      case 14: CHK(conv_tables_sub<14>(c, q, psi, 0, 0, &pl->pd[0])); break;
      case 15: CHK(conv_tables_sub<15>(c, q, psi, 0, 0, &pl->pd[0])); break;
    }
  } else if (pl->split >= 8) {
    const int LS = pl->split == 16 ? 4 : 3, R = pl->split;
    for (unsigned g = 0; g < (unsigned)R; g++)
      CHK(conv_tables_sub<15>(c, q, psi, LS, g, &pl->pd[g]));
    auto mk = [&](uint64_t w) {
      TW t;
      t.w = w;
      t.wp = hxh::shoup(w, q);
      return t;
    };
    const uint64_t rinv = hxh::invmod((uint64_t)R % q, q);
    auto fill = [&](auto& S) {
      for (unsigned idx = 1; idx < (unsigned)R; idx++) {
        const uint64_t T = hxh::powmod(psi, hx::brev_bits(idx, logn), q);
        S.T[idx] = mk(T);
        S.iT[idx] = mk(hxh::invmod(T, q));
      }
      S.T[0] = S.iT[0] = mk(0);
      S.inv = mk(rinv);
      S.iT1e = mk(hxh::mulmod(S.iT[1].w, rinv, q));
    };
    if (R == 16)
      fill(h.S16);
    else
      fill(h.S8);
  } else {
    for (unsigned g = 0; g < 4; g++) {
      if (logn == 16)
        CHK(conv_tables_sub<14>(c, q, psi, 2, g, &pl->pd[g]));
      else
        CHK(conv_tables_sub<15>(c, q, psi, 2, g, &pl->pd[g]));
    }
    auto mk = [&](uint64_t w) {
      TW t;
      t.w = w;
      t.wp = hxh::shoup(w, q);
      return t;
    };
    auto prev = [&](unsigned idx) { return hxh::powmod(psi, hx::brev_bits(idx, logn), q); };
    uint64_t T1 = prev(1), T2 = prev(2), T3 = prev(3), quarter = hxh::invmod(4 % q, q);
    h.S.T1 = mk(T1);
    h.S.T2 = mk(T2);
    h.S.T3 = mk(T3);
    h.S.iT2 = mk(hxh::invmod(T2, q));
    h.S.iT3 = mk(hxh::invmod(T3, q));
    h.S.iT1q = mk(hxh::mulmod(hxh::invmod(T1, q), quarter, q));
    h.S.quarter = mk(quarter);
  }
  HIPCHK(hipMalloc((void**)&pl->dev, sizeof h));
  HIPCHK(hipMemcpy(pl->dev, &h, sizeof h, hipMemcpyHostToDevice));
  return HX_OK;
}

static dim3 grid2(uint32_t n, size_t segs)
{
  unsigned bx = (n + 255) / 256;
  if (bx > 64)
    bx = 64;
  if (bx < 1)
    bx = 1;
  return dim3(bx, (unsigned)segs);
}

// In-place convolution with a precomputed transform: buf[(ri*batch+b)][2^logn] (time domain)
// <- buf * hat_ri.  hats[ri] == nullptr for all rows: forward transform only (result left in
// transform order in buf, or in qbuf for split sizes).  plans[ri]: the row's transform plan.
static int conv_core(hx_ctx* c, uint64_t* buf, uint64_t* qbuf, const std::vector<const ConvPlan*>& plans,
                     const std::vector<const uint64_t*>& hatv, int batch)
{
  const int R = (int)plans.size();
  if (R == 0)
    return HX_OK;
  if (R > MAX_ROWS / 4)
    return fail(HX_ERR_INVALID, "internal: conv chunk too large");
  const ConvPlan& p0 = *plans[0];
  const int logn = p0.logn;
  const int split = p0.split;
  if (split && R > MAX_ROWS / split)
    return fail(HX_ERR_INVALID, "internal: conv chunk too large");
  const uint32_t N = 1u << logn;
  const bool fwd_only = hatv[0] == nullptr;
  hx::PtrList cps, hats;
  for (int r = 0; r < R; r++) {
    cps.p[r] = plans[r]->dev;
    hats.p[r] = hatv[r];
  }
  std::vector<std::pair<int, int>> rows;
  if (!split) {
    for (int r = 0; r < R; r++)
      rows.emplace_back(r, plans[r]->pd[0]);
    CHK(ntt_launch(c, logn, c->d_cprimes, buf, buf, rows, batch, false));
    if (fwd_only)
      return HX_OK;
    HX_LAUNCH(hx::conv_pointwise_kernel, grid2(N, (size_t)R * batch), dim3(256), 0, c->stream,
                       buf, hats, cps, 1, batch, N);
    HIPCHK(hipGetLastError());
    return ntt_launch(c, logn, c->d_cprimes, buf, buf, rows, batch, true);
  }
  const uint32_t Q = N / (uint32_t)split;
  const int lsub = logn - (split == 16 ? 4 : (split == 8 ? 3 : 2));
  for (int r = 0; r < R; r++)
    for (int g = 0; g < split; g++)
      rows.emplace_back(r * split + g, plans[r]->pd[g]);
  auto split_launch = [&](int inverse) {
    if (split == 16)
      HX_LAUNCH(hx::conv_splitN_kernel<4>, grid2(Q, (size_t)R * batch), dim3(256), 0, c->stream, buf, qbuf,
                         cps, batch, Q, inverse);
    else if (split == 8)
      HX_LAUNCH(hx::conv_splitN_kernel<3>, grid2(Q, (size_t)R * batch), dim3(256), 0, c->stream, buf, qbuf,
                         cps, batch, Q, inverse);
    else
      HX_LAUNCH(hx::conv_split_kernel, grid2(Q, (size_t)R * batch), dim3(256), 0, c->stream, buf, qbuf,
                         cps, batch, Q, inverse);
    return hipGetLastError();
  };
  HIPCHK(split_launch(0));
  CHK(ntt_launch(c, lsub, c->d_cprimes, qbuf, qbuf, rows, batch, false));
  if (fwd_only)
    return HX_OK;
  HX_LAUNCH(hx::conv_pointwise_kernel, grid2(Q, (size_t)R * split * batch), dim3(256), 0, c->stream,
                     qbuf, hats, cps, split, batch, Q);
  HIPCHK(hipGetLastError());
  CHK(ntt_launch(c, lsub, c->d_cprimes, qbuf, qbuf, rows, batch, true));
  HIPCHK(split_launch(1));
  return HX_OK;
}

// one row (batch blocks of N words at `sub`) of an aux-mode prime: three convolutions modulo the
// auxiliary primes, then the exact recombination modulo q
static int aux_conv_row(hx_ctx* c, uint64_t* sub, const BluePrime* bp, int batch, int which, int hat_sel)
{
  const int logn = bp->aconv[0][which].logn;
  const size_t n = (size_t)batch << logn;
  CHK(ensure_scratch(c, 8, 3 * n));
  CHK(ensure_scratch(c, 9, n));
  const unsigned blocks = (unsigned)std::min<size_t>(4096, (n + 255) / 256);
  for (int j = 0; j < 3; j++) {
    uint64_t* t = c->scratch[8] + (size_t)j * n;
    HX_LAUNCH(hx::aux_load_kernel, dim3(blocks), dim3(256), 0, c->stream, sub, t, n, bp->crt.A[j]);
    HIPCHK(hipGetLastError());
    std::vector<const ConvPlan*> pl(1, &bp->aconv[j][which]);
    std::vector<const uint64_t*> hv(1, bp->d_ahat[j][hat_sel]);
    CHK(conv_core(c, t, c->scratch[9], pl, hv, batch));
  }
  HX_LAUNCH(hx::crt3_kernel, dim3(blocks), dim3(256), 0, c->stream, c->scratch[8], c->scratch[8] + n,
                     c->scratch[8] + 2 * n, sub, n, bp->crt);
  HIPCHK(hipGetLastError());
  return HX_OK;
}

// prime_of_row[ri] = context prime index; which = 0/1/2 (bk/n1/n2); hat_sel: the hat table
static int conv_apply(hx_ctx* c, uint64_t* buf, uint64_t* qbuf, const std::vector<int>& prime_of_row,
                      int batch, int which, int hat_sel)
{
  const int R = (int)prime_of_row.size();
  bool any_aux = false;
  for (int r = 0; r < R; r++)
    any_aux = any_aux || c->blue[prime_of_row[r]]->aux;
  if (!any_aux) {
    std::vector<const ConvPlan*> pl(R);
    std::vector<const uint64_t*> hv(R);
    for (int r = 0; r < R; r++) {
      pl[r] = &c->blue[prime_of_row[r]]->conv[which];
      hv[r] = hat_sel >= 0 ? c->blue[prime_of_row[r]]->d_hat[hat_sel] : nullptr;
    }
    return conv_core(c, buf, qbuf, pl, hv, batch);
  }
  // a chunk with aux-mode primes (exotic: e.g. the reference's legacy fixture): row by row
  for (int r = 0; r < R; r++) {
    const BluePrime* bp = c->blue[prime_of_row[r]];
    const int logn = bp->aux ? bp->aconv[0][which].logn : bp->conv[which].logn;
    uint64_t* sub = buf + ((size_t)r * batch << logn);
    if (bp->aux) {
      CHK(aux_conv_row(c, sub, bp, batch, which, hat_sel));
    } else {
      std::vector<const ConvPlan*> pl(1, &bp->conv[which]);
      std::vector<const uint64_t*> hv(1, bp->d_hat[hat_sel]);
      CHK(conv_core(c, sub, qbuf, pl, hv, batch));
    }
  }
  return HX_OK;
}

// upload a time-domain polynomial (host, length <= 2^logn), transform it on the device with the
// prime's own plan and keep the result as hat table `slot`
static int make_hat(hx_ctx* c, int prime, int which, int slot, const std::vector<uint64_t>& poly)
{
  BluePrime* bp = c->blue[prime];
  if (bp->aux) {  // the same polynomial reduced and transformed modulo each auxiliary prime
    const int logn = bp->aconv[0][which].logn;
    const size_t N = (size_t)1 << logn;
    CHK(ensure_scratch(c, 4, N));
    CHK(ensure_scratch(c, 5, N));
    for (int j = 0; j < 3; j++) {
      std::vector<uint64_t> h(N, 0);
      for (size_t i = 0; i < poly.size(); i++)
        h[i] = poly[i] % bp->crt.A[j];
      HIPCHK(hipMemcpyAsync(c->scratch[4], h.data(), N * 8, hipMemcpyHostToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
      std::vector<const ConvPlan*> pl(1, &bp->aconv[j][which]);
      std::vector<const uint64_t*> hv(1, nullptr);
      CHK(conv_core(c, c->scratch[4], c->scratch[5], pl, hv, 1));
      HIPCHK(hipMalloc((void**)&bp->d_ahat[j][slot], N * 8));
      HIPCHK(hipMemcpyAsync(bp->d_ahat[j][slot], bp->aconv[j][which].split ? c->scratch[5] : c->scratch[4], N * 8,
                            hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    return HX_OK;
  }
  const int logn = bp->conv[which].logn;
  const size_t N = (size_t)1 << logn;
  CHK(ensure_scratch(c, 4, N));
  CHK(ensure_scratch(c, 5, N));
  std::vector<uint64_t> h(N, 0);
  std::copy(poly.begin(), poly.end(), h.begin());
  HIPCHK(hipMemcpyAsync(c->scratch[4], h.data(), N * 8, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  std::vector<int> pr(1, prime);
  CHK(conv_apply(c, c->scratch[4], c->scratch[5], pr, 1, which, -1));
  HIPCHK(hipMalloc((void**)&bp->d_hat[slot], N * 8));
  HIPCHK(hipMemcpyAsync(bp->d_hat[slot], bp->conv[which].split ? c->scratch[5] : c->scratch[4], N * 8,
                        hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  {
    // the same transform as Shoup pairs {w, floor(w 2^64 / q)}: what the fused convolution kernel multiplies by
    std::vector<uint64_t> w(N);
    HIPCHK(hipMemcpy(w.data(), bp->d_hat[slot], N * 8, hipMemcpyDeviceToHost));
    std::vector<TW> t(N);
    const uint64_t q = c->primes[prime].q;
    for (size_t i = 0; i < N; i++) {
      t[i].w = w[i];
      t[i].wp = hxh::shoup(w[i], q);
    }
    HIPCHK(hipMalloc((void**)&bp->d_hatw[slot], N * sizeof(TW)));
    HIPCHK(hipMemcpy(bp->d_hatw[slot], t.data(), N * sizeof(TW), hipMemcpyHostToDevice));
  }
  return HX_OK;
}

// Cmodulus constructor, general-m branch (src/CModulus.cpp:140-181) + BluesteinInit x2
static int blue_prime_create(hx_ctx* c, int idx)
{
  const PrimeHost& ph = c->primes[idx];
  const uint64_t q = ph.q, m = c->m;
  if (c->bk == 0) {
    c->bk = next_pow2_exp(2 * m - 1);
    c->dq = (uint32_t)(m - 1 - c->phim);
    c->n1 = std::max(1, next_pow2_exp(2 * (uint64_t)c->dq + 1));
    c->n2 = std::max(1, next_pow2_exp(m));
    // Q Phi_m has degree m - 1 but only its low phi(m) coefficients are unknown: the ones above are x's own (the
    // remainder has degree < phi(m)).  Modulo X^N3 + 1 with phi(m) <= N3 and m - 1 < 2 N3 coefficient i of the
    // negacyclic product is (Q Phi)_i - (Q Phi)_(i + N3) = (Q Phi)_i - x_(i + N3): half the transform size.
    {
      const int n3 = std::max(1, next_pow2_exp(c->phim));
      // (Q itself, degree <= dq, must fit the transform unwrapped: dq < N3, which with phi(m) <= N3 gives m - 1 < 2 N3)
      c->n3 = (n3 < c->n2 && (uint64_t)c->dq < ((uint64_t)1 << n3)) ? n3 : c->n2;
    }
    c->mpad = (uint32_t)((m + 1) & ~(uint64_t)1);
    c->phi_coef = cyclo_product(m, c->phim + 1, true);
    std::vector<int64_t> psi = cyclo_product(m, c->dq + 2, false);
    c->psi_low.assign(psi.begin(), psi.begin() + c->dq + 1);
  }
  int maxk = std::max(c->bk, std::max(c->n1, c->n2));
  if (maxk > 19)
    return fail(HX_ERR_UNSUPPORTED, "m too large for the Bluestein path (conv size 2^%d)", maxk);
  if ((int)c->blue.size() <= idx)
    c->blue.resize(idx + 1, nullptr);
  BluePrime* bp = new BluePrime();
  c->blue[idx] = bp;
  const int sizes[3] = {c->bk, c->n1, c->n2};
  if ((q - 1) % ((uint64_t)1 << (maxk + 1)) == 0) {
    // PrimeGenerator primes: q = 2^k t m + 1 with a large k -- the convolutions run modulo q itself
    const uint64_t gen = hxh::find_prim_root(q, (uint64_t)1 << (maxk + 1));
    for (int w = 0; w < 3; w++) {
      uint64_t psi = hxh::powmod(gen, (uint64_t)1 << (maxk - sizes[w]), q);
      CHK(conv_plan_create(c, q, sizes[w], psi, &bp->conv[w]));
    }
    if (c->n3 < c->n2)
      CHK(conv_plan_create(c, q, c->n3, hxh::powmod(gen, (uint64_t)1 << (maxk - c->n3), q), &bp->conv[3]));
  } else {
    // no 2^(maxk+1)-th root of unity modulo q (e.g. the primes of the reference's legacy fixture):
    // exact convolutions through three auxiliary NTT primes
    bp->aux = true;
    if (c->aux_q[0] == 0) {
      int found = 0;
      for (uint64_t a = ((uint64_t)1 << 60) - ((uint64_t)1 << 20) + 1; found < 3 && a > ((uint64_t)1 << 59);
           a -= (uint64_t)1 << 20)
        if (hxh::is_prime(a))
          c->aux_q[found++] = a;
      if (found < 3)
        return fail(HX_ERR_UNSUPPORTED, "internal: no auxiliary NTT primes");
    }
    hx::Crt3Dev& C = bp->crt;
    memset(&C, 0, sizeof C);
    for (int j = 0; j < 3; j++) {
      const uint64_t A = c->aux_q[j];
      C.A[j] = A;
      C.muA[j] = (uint64_t)((((hxh::u128)1) << 120) / A);
      const uint64_t gen = hxh::find_prim_root(A, (uint64_t)1 << (maxk + 1));
      for (int w = 0; w < 3; w++) {
        uint64_t psi = hxh::powmod(gen, (uint64_t)1 << (maxk - sizes[w]), A);
        CHK(conv_plan_create(c, A, sizes[w], psi, &bp->aconv[j][w]));
      }
    }
    C.inv01 = hxh::invmod(C.A[0] % C.A[1], C.A[1]);
    C.a0m2 = C.A[0] % C.A[2];
    C.inv012 = hxh::invmod(hxh::mulmod(C.A[0] % C.A[2], C.A[1] % C.A[2], C.A[2]), C.A[2]);
    C.q = q;
    C.k = (uint32_t)hxh::bitlen(q);
    C.mu = (uint64_t)((((hxh::u128)1) << (2 * C.k)) / q);
    C.mu64 = (uint64_t)((((hxh::u128)1) << 64) / q);
    C.a0q = C.A[0] % q;
    C.a01q = hxh::mulmod(C.A[0] % q, C.A[1] % q, q);
  }
  // powers[i] = root^(i^2 mod e), chirp b (src/bluestein.cpp:76-132)
  const uint64_t e = (m % 2 == 0) ? 2 * m : m;
  std::vector<TW> pw(m), ipw(m);
  const size_t NB = (size_t)1 << c->bk;
  std::vector<uint64_t> b(NB, 0), ib(NB, 0);
  for (uint64_t i = 0; i < m; i++) {
    uint64_t isq = hxh::mulmod(i, i, e);
    uint64_t v = hxh::powmod(ph.root, isq, q), iv = hxh::powmod(ph.rinv, isq, q);
    pw[i].w = v;
    pw[i].wp = hxh::shoup(v, q);
    ipw[i].w = iv;
    ipw[i].wp = hxh::shoup(iv, q);
    // forward chirp uses rInv^(i^2) = iv, the inverse-direction chirp uses root^(i^2) = v
    if (m == e) {
      b[i] = iv;
      ib[i] = v;
    } else {
      b[m - 1 + i] = iv;
      b[m - 1 - i] = iv;
      ib[m - 1 + i] = v;
      ib[m - 1 - i] = v;
    }
  }
  HIPCHK(hipMalloc((void**)&bp->d_powers, sizeof(TW) * m));
  HIPCHK(hipMalloc((void**)&bp->d_ipowers, sizeof(TW) * m));
  HIPCHK(hipMemcpy(bp->d_powers, pw.data(), sizeof(TW) * m, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(bp->d_ipowers, ipw.data(), sizeof(TW) * m, hipMemcpyHostToDevice));
  hx::BluePrimeDev hd;
  hd.q = q;
  hd.powers = bp->d_powers;
  hd.ipowers = bp->d_ipowers;
  uint64_t minv = hxh::invmod(m % q, q);
  hd.minv.w = minv;
  hd.minv.wp = hxh::shoup(minv, q);
  HIPCHK(hipMalloc((void**)&bp->dev, sizeof hd));
  HIPCHK(hipMemcpy(bp->dev, &hd, sizeof hd, hipMemcpyHostToDevice));
  CHK(make_hat(c, idx, 0, 0, b));
  CHK(make_hat(c, idx, 0, 1, ib));
  // rem Phi_m helpers: -Psi mod X^(dq+1) and Phi_m, reduced mod q
  std::vector<uint64_t> npsi(c->dq + 1), phi(c->phim + 1);
  for (size_t i = 0; i < npsi.size(); i++) {
    int64_t v = -c->psi_low[i];
    npsi[i] = v >= 0 ? (uint64_t)v % q : q - ((uint64_t)(-v) % q);
    if (npsi[i] == q)
      npsi[i] = 0;
  }
  for (size_t i = 0; i < phi.size(); i++) {
    int64_t v = c->phi_coef[i];
    phi[i] = v >= 0 ? (uint64_t)v % q : q - ((uint64_t)(-v) % q);
    if (phi[i] == q)
      phi[i] = 0;
  }
  CHK(make_hat(c, idx, 1, 2, npsi));
  CHK(make_hat(c, idx, 2, 3, phi));
  if (!bp->aux && c->n3 < c->n2) {   // Phi_m modulo X^N3 + 1: the coefficients from N3 on wrap with a minus sign
    const size_t N3 = (size_t)1 << c->n3;
    std::vector<uint64_t> folded(N3, 0);
    for (size_t i = 0; i < phi.size(); i++) {
      uint64_t& f = folded[i % N3];
      f = ((i / N3) & 1) ? (f >= phi[i] ? f - phi[i] : f + q - phi[i]) : (f + phi[i] >= q ? f + phi[i] - q : f + phi[i]);
    }
    CHK(make_hat(c, idx, 3, 4, folded));
  }
  // m = 5 * 17 * 257 on a prime with the 256-th roots of unity: the transform itself runs as Good-Thomas x Rader
  // (pfa_core.h), ONE launch per direction -- the Proth-form Montgomery product on rows of such primes, the generic
  // one on the others (the 40-bit small primes of a chain; every row under HX_NO_PROTH); the tables above stay for
  // HX_NO_PFA / HX_PFA_NO_REM
  if (!bp->aux && !c->sw.no_pfa && hx::pfa::host::supported(m, q)) {
    if (!c->d_pfa_idx) {
      std::vector<uint16_t> ix(16384 + 258 + 256, 0);
      hx::pfa::host::build_index_tables(ix.data(), ix.data() + 16384, ix.data() + 16384 + 258);
      HIPCHK(hipMalloc((void**)&c->d_pfa_idx, ix.size() * sizeof(uint16_t)));
      HIPCHK(hipMemcpy(c->d_pfa_idx, ix.data(), ix.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    std::vector<uint64_t> tab(hx::pfa::TAB_WORDS);
    hx::pfa::host::build_prime_table(q, ph.root, tab.data());
    HIPCHK(hipMalloc((void**)&bp->d_pfa, tab.size() * 8));
    HIPCHK(hipMemcpy(bp->d_pfa, tab.data(), tab.size() * 8, hipMemcpyHostToDevice));
  }
  return HX_OK;
}

// Cmodulus::FFT_aux / iFFT, general-m branch, for the listed rows of a [rows][batch][phi] buffer
static int bluestein_rows(hx_ctx* c, const uint64_t* in, uint64_t* out,
                          const std::vector<std::pair<int, int>>& rows, int batch, bool inverse)
{
  const uint32_t phim = c->phim, m = (uint32_t)c->m;
  const uint32_t NB = 1u << c->bk, N1 = 1u << c->n1, N2 = 1u << c->n2;
  {
    // a list that mixes rows with and without the Good-Thomas x Rader tables (m = 21845: the 40-bit small primes of a
    // chain are not of the Proth form) is served in two parts -- one such row used to send the whole launch, the 60-bit
    // rows included, through Bluestein: 27 % of a multiply at m = 21845, bits = 950 (tools/prof_config5_ring.py)
    // (and the rows with tables by the arithmetic their prime takes -- Proth-form or generic Montgomery product: one
    // per launch, pfa_kernels.hip)
    std::vector<std::pair<int, int>> part[3];   // tables + Proth form, tables + other prime, no tables
    for (auto& rp : rows) {
      const bool has = rp.second >= 0 && rp.second < (int)c->blue.size() && c->blue[rp.second] && c->blue[rp.second]->d_pfa;
      // (PrimeHost::proth describes the power-of-two row tables only: decided here from the prime itself)
      part[has ? ((hx::is_proth32(c->primes[rp.second].q) && !c->sw.no_proth) ? 0 : 1) : 2].push_back(rp);
    }
    const int nparts = (int)!part[0].empty() + (int)!part[1].empty() + (int)!part[2].empty();
    if (nparts > 1) {
      for (auto& pr : part)
        if (!pr.empty())
          CHK(bluestein_rows(c, in, out, pr, batch, inverse));
      return HX_OK;
    }
  }
  // chunk so that the convolution buffers stay below ~1 GiB each
  size_t per_row = (size_t)batch * NB * 8;
  int chunk = (int)std::max<size_t>(1, std::min<size_t>(MAX_ROWS / (c->bk > 18 ? 16 : (c->bk > 17 ? 8 : 4)), ((size_t)1 << 30) / per_row));
  // fused path (conv_dev.h): every convolution is ONE launch of the convolution row kernel + one
  // element-wise pass; needs primes that convolve modulo themselves and sizes the row kernels take
  // (2^13..2^15, the chirp convolution optionally as a radix-4 split)
  const bool old_path = c->sw.blue_old;
  auto sub_ok = [](const ConvPlan& pl) {
    const int l = pl.split == 4 ? pl.logn - 2 : pl.logn;
    return (pl.split == 0 || pl.split == 4) && l >= 13 && l <= 15;
  };
  bool fused = !old_path;
  for (auto& rp : rows) {
    if (rp.second >= (int)c->blue.size() || !c->blue[rp.second])
      return fail(HX_ERR_INVALID, "prime %d has no Bluestein tables", rp.second);
    const BluePrime* b0 = c->blue[rp.second];
    fused = fused && !b0->aux && sub_ok(b0->conv[0]) &&
            (!inverse || (b0->conv[1].split == 0 && sub_ok(b0->conv[1]) && b0->conv[2].split == 0 && sub_ok(b0->conv[2])));
  }
  if (fused)
    chunk = std::min(chunk, hx::CONV_MAXROWS);
  // every row on a prime that has the Good-Thomas x Rader tables (m = 21845, blue_prime_create)
  bool pfa = fused;
  for (auto& rp : rows)
    pfa = pfa && c->blue[rp.second]->d_pfa != nullptr;
  for (size_t first = 0; fused && first < rows.size(); first += chunk) {
    const int R = (int)std::min<size_t>(chunk, rows.size() - first);
    const size_t segs = (size_t)R * batch;
    const BluePrime* b0 = c->blue[rows[first].second];
    const int split = b0->conv[0].split ? 4 : 1, lsub = b0->conv[0].logn - (split == 4 ? 2 : 0);
    NttRows nr;
    hx::PtrList bp, cps;
    hx::ConvRows CR;
    memset(&CR, 0, sizeof CR);
    for (int r = 0; r < R; r++) {
      const BluePrime* b = c->blue[rows[first + r].second];
      nr.row[r] = (uint16_t)rows[first + r].first;
      nr.prime[r] = (uint16_t)rows[first + r].second;
      bp.p[r] = b->dev;
      cps.p[r] = b->conv[0].dev;
      CR.bp[r] = b->dev;
      CR.row[r] = (uint16_t)rows[first + r].first;
    }
    auto set_conv = [&](int which, int slot, int sp) {
      for (int r = 0; r < R; r++) {
        const BluePrime* b = c->blue[rows[first + r].second];
        CR.hat[r] = b->d_hatw[slot];
        CR.cp[r] = b->conv[which].dev;
        for (int g = 0; g < sp; g++)
          CR.pd[r * sp + g] = (uint16_t)b->conv[which].pd[g];
      }
    };
    hx::ConvRowArgs A;
    memset(&A, 0, sizeof A);
    A.batch = (uint32_t)batch;
    A.phim = phim;
    A.m = m;
    A.zidx = c->d_zms_index;
    hipError_t e = hipSuccess;
    uint64_t* xfull = nullptr;
    if (inverse && !(pfa && !c->sw.pfa_no_rem)) {
      CHK(ensure_scratch(c, 6, segs * c->mpad));
      CHK(ensure_scratch(c, 7, segs * N1));
      xfull = c->scratch[6];
    }
    if (pfa) {
      // Good-Thomas x Rader (pfa_core.h): the whole forward transform, or the inverse one up to X (all m words)
      hx::PfaRows PR;
      memset(&PR, 0, sizeof PR);
      for (int r = 0; r < R; r++) {
        PR.tab[r] = c->blue[rows[first + r].second]->d_pfa;
        PR.row[r] = (uint16_t)rows[first + r].first;
        PR.prime[r] = (uint16_t)rows[first + r].second;
      }
      // (the inverse with rem Phi_m and 1/m behind it in the same launch -- binomial passes, no multiplication --
      // unless HX_PFA_NO_REM keeps them on the convolution kernels below)
      const int mode = !inverse ? 0 : (c->sw.pfa_no_rem ? 1 : 2);
      e = hx::launch_pfa_rows(mode, hx::is_proth32(c->primes[rows[first].second].q) && !c->sw.no_proth, in, mode == 1 ? xfull : out, PR, R, c->d_primes, c->d_pfa_idx, c->d_pfa_idx + 16384,
                              c->d_pfa_idx + 16384 + 258, batch, c->mpad, c->stream);
      if (e != hipSuccess)
        return fail(HX_ERR_DEVICE, "Good-Thomas x Rader launch failed: %s", hipGetErrorString(e));
      if (mode != 1)
        continue;
    } else {
    CHK(ensure_scratch(c, 5, segs * NB));
    uint64_t* qbuf = c->scratch[5];
    A.in = in;
    A.out = qbuf;
    A.split = (uint32_t)split;
    A.dst_mode = hx::CONV_DST_SUB;
    A.src_mode = inverse ? hx::CONV_SRC_SCATTER : hx::CONV_SRC_BLUE_PRE;
    set_conv(0, inverse ? 1 : 0, split);
    e = hx::launch_conv_rows(lsub, A, CR, R * split, c->d_cprimes, c->d_tw, c->stream);
    if (e != hipSuccess)
      return fail(HX_ERR_DEVICE, "convolution launch failed: %s", hipGetErrorString(e));
    // window / fold + second twist (+ the inverse split when the result is in four sub-blocks)
    uint64_t* pdst = inverse ? xfull : out;
    const uint32_t pn = inverse ? m : phim;
    if (split == 4)
      HX_LAUNCH(hx::blue_post4_kernel, grid2(pn, segs), dim3(256), 0, c->stream, qbuf, pdst, nr, bp, cps, batch, phim, m,
                (uint32_t)lsub, c->mpad, c->d_zms, inverse ? 0 : 1);
    else
      HX_LAUNCH(hx::blue_post_kernel, grid2(pn, segs), dim3(256), 0, c->stream, qbuf, pdst, nr, bp, batch, phim, m, NB,
                c->mpad, c->d_zms, inverse ? 0 : 1);
    HIPCHK(hipGetLastError());
    if (!inverse)
      continue;
    }
    // rem Phi_m: Q = rev_d( top(x) * (-Psi) mod X^(d+1) ), r = x - Q*Phi_m, then * m^-1
    uint64_t* wbuf = c->scratch[7];
    A.split = 1;
    A.src_mode = hx::CONV_SRC_REV;
    A.in = xfull;
    A.in_stride = c->mpad;
    A.base = m - 1;
    A.d = c->dq;
    A.out = wbuf;
    set_conv(1, 2, 1);
    e = hx::launch_conv_rows(c->n1, A, CR, R, c->d_cprimes, c->d_tw, c->stream);
    if (e != hipSuccess)
      return fail(HX_ERR_DEVICE, "convolution launch failed: %s", hipGetErrorString(e));
    A.in = wbuf;
    A.in_stride = N1;
    A.base = c->dq;
    A.aux = xfull;
    A.aux_stride = c->mpad;
    A.dst_mode = hx::CONV_DST_FINAL;
    A.out = out;
    // Q Phi_m at half the size where the aliased coefficients are x's own (n3 < n2, blue_prime_create)
    const bool small = c->n3 < c->n2 && b0->conv[3].dev && b0->conv[3].split == 0 && sub_ok(b0->conv[3]);
    A.alias = small ? 1u : 0u;
    set_conv(small ? 3 : 2, small ? 4 : 3, 1);
    e = hx::launch_conv_rows(small ? c->n3 : c->n2, A, CR, R, c->d_cprimes, c->d_tw, c->stream);
    if (e != hipSuccess)
      return fail(HX_ERR_DEVICE, "convolution launch failed: %s", hipGetErrorString(e));
  }
  if (fused)
    return HX_OK;
  for (size_t first = 0; first < rows.size(); first += chunk) {
    const int R = (int)std::min<size_t>(chunk, rows.size() - first);
    NttRows nr;
    hx::PtrList bp;
    std::vector<int> pr(R);
    for (int r = 0; r < R; r++) {
      nr.row[r] = (uint16_t)rows[first + r].first;
      nr.prime[r] = (uint16_t)rows[first + r].second;
      pr[r] = rows[first + r].second;
      if (pr[r] >= (int)c->blue.size() || !c->blue[pr[r]])
        return fail(HX_ERR_INVALID, "prime %d has no Bluestein tables", pr[r]);
      bp.p[r] = c->blue[pr[r]]->dev;
    }
    const size_t segs = (size_t)R * batch;
    CHK(ensure_scratch(c, 4, segs * NB));
    CHK(ensure_scratch(c, 5, segs * NB));
    uint64_t* cbuf = c->scratch[4];
    uint64_t* qbuf = c->scratch[5];
    if (!inverse) {
      HX_LAUNCH(hx::blue_pre_kernel, grid2(NB, segs), dim3(256), 0, c->stream, in, cbuf, nr, bp,
                         batch, phim, NB, 0);
      HIPCHK(hipGetLastError());
      CHK(conv_apply(c, cbuf, qbuf, pr, batch, 0, 0));
      HX_LAUNCH(hx::blue_post_kernel, grid2(phim, segs), dim3(256), 0, c->stream, cbuf, out, nr,
                         bp, batch, phim, m, NB, c->mpad, c->d_zms, 1);
      HIPCHK(hipGetLastError());
      continue;
    }
    CHK(ensure_scratch(c, 6, segs * c->mpad));
    CHK(ensure_scratch(c, 7, segs * std::max(N1, N2)));
    uint64_t* xfull = c->scratch[6];
    uint64_t* wbuf = c->scratch[7];
    HX_LAUNCH(hx::blue_scatter_kernel, grid2(NB, segs), dim3(256), 0, c->stream, in, cbuf, nr, bp,
                       batch, phim, m, NB, c->d_zms_index);
    HIPCHK(hipGetLastError());
    CHK(conv_apply(c, cbuf, qbuf, pr, batch, 0, 1));
    HX_LAUNCH(hx::blue_post_kernel, grid2(m, segs), dim3(256), 0, c->stream, cbuf, xfull, nr, bp,
                       batch, phim, m, NB, c->mpad, c->d_zms, 0);
    HIPCHK(hipGetLastError());
    // rem Phi_m: Q = rev_d( top(x) * (-Psi) mod X^(d+1) ), r = x - Q*Phi_m
    HX_LAUNCH(hx::blue_rev_kernel, grid2(N1, segs), dim3(256), 0, c->stream, xfull, wbuf, batch,
                       c->mpad, m - 1, c->dq, N1);
    HIPCHK(hipGetLastError());
    CHK(conv_apply(c, wbuf, qbuf, pr, batch, 1, 2));
    // Q (zero padded to N2) goes to cbuf, which is free again
    HX_LAUNCH(hx::blue_rev_kernel, grid2(N2, segs), dim3(256), 0, c->stream, wbuf, cbuf, batch, N1,
                       c->dq, c->dq, N2);
    HIPCHK(hipGetLastError());
    CHK(conv_apply(c, cbuf, qbuf, pr, batch, 2, 3));
    HX_LAUNCH(hx::blue_final_kernel, grid2(phim, segs), dim3(256), 0, c->stream, xfull, cbuf, out,
                       nr, bp, batch, phim, c->mpad, N2);
    HIPCHK(hipGetLastError());
  }
  return HX_OK;
}

extern "C" int hx_ctx_add_prime(hx_ctx* c, uint64_t q, uint64_t root, int* idx_out)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null context");
  CTX_ENTER(c);
  if (c->capturing)
    return fail(HX_ERR_INVALID, "hx_ctx_add_prime while a graph is being captured");
  // 16q <= 2^64 is what the lazy butterflies of the row kernels need (ntt_core.h); the reference
  // cannot make larger primes either (HELIB_SP_NBITS <= 60, src/PrimeGenerator.h:54-59)
  if (q < 3 || q >= (1ull << 60) || !hxh::is_prime(q))
    return fail(HX_ERR_INVALID, "q=%llu is not a prime below 2^60", (unsigned long long)q);
  if ((int)c->primes.size() >= c->primes_cap)
    return fail(HX_ERR_UNSUPPORTED, "too many primes");
  uint64_t e = c->pow2 ? c->m : ((c->m % 2 == 0) ? 2 * c->m : c->m);
  if ((q - 1) % e != 0)
    return fail(HX_ERR_INVALID, "e=%llu does not divide q-1 (no primitive root)",
                (unsigned long long)e);
  if (root == 0)
    root = hxh::find_prim_root(q, e);
  // verify the order (reference: FindPrimRootT's independent check)
  // (order exactly e: root^e = 1 and root^(e/p) != 1 for every prime factor p of e)
  bool prim = hxh::powmod(root, e, q) == 1;
  {
    uint64_t rest = e;
    for (uint64_t p = 2; prim && p * p <= rest; p++)
      if (rest % p == 0) {
        prim = hxh::powmod(root, e / p, q) != 1;
        while (rest % p == 0)
          rest /= p;
      }
    if (prim && rest > 1)
      prim = hxh::powmod(root, e / rest, q) != 1;
  }
  if (!prim)
    return fail(HX_ERR_INVALID, "root is not a primitive %llu-th root of unity mod q",
                (unsigned long long)e);
  PrimeHost ph;
  ph.q = q;
  ph.root = root;
  ph.rinv = hxh::invmod(root, q);
  if (c->pow2) {
    switch (c->logn) {
      case 13: CHK(upload_tw<13>(c, ph)); break;
      case 14: CHK(upload_tw<14>(c, ph)); break;
      case 15:
        CHK(upload_tw<15>(c, ph));
        if (c->sw.half15) {
          // the forward transform as two 2^14-point workgroups per row (ntt_kernels.hip ntt_row_half15_kernel): the
          // sub-transforms' tables (what a split power-of-two ring builds with OUT = 1) and the first stage's twiddle
          int i0 = -1, i1 = -1;
          CHK(conv_tables_sub<14>(c, q, root, 1, 0, &i0));
          CHK(conv_tables_sub<14>(c, q, root, 1, 1, &i1));
          if (i1 != i0 + 1)
            return fail(HX_ERR_DEVICE, "internal: half-row table entries are not adjacent");
          const uint64_t t1 = hxh::powmod(root, (uint64_t)1 << 14, q);   // psi_rev[1] = psi^(N/2)
          const uint64_t h3[3] = {t1, hxh::shoup(t1, q), hx::tw_mont_form(t1, q)};
          for (int g = 0; g < 2; g++)
            HIPCHK(hipMemcpy(reinterpret_cast<char*>(c->d_cprimes + i0 + g) + offsetof(PrimeDev, half_t1), h3, sizeof h3,
                             hipMemcpyHostToDevice));
          ph.half_pd = i0;
        }
        break;
      default:
        if (c->logn >= 1 && c->logn <= 12)
          CHK(upload_tw_small(c, ph));
        break;  // N = 2^16..2^18: split plans below; beyond that element-wise / RNS ops still work
    }
  }
  ConvPlan big;
  if (c->pow2 && c->logn >= 16 && c->logn <= 19)
    CHK(conv_plan_create(c, q, c->logn, root, &big));
  PrimeDev pd;
  memset(&pd, 0, sizeof pd);
  pd.q = q;
  pd.q2 = 2 * q;
  pd.k = (uint32_t)hxh::bitlen(q);
  pd.mu = (uint64_t)((((hxh::u128)1) << (2 * pd.k)) / q);
  pd.mu64 = (uint64_t)((((hxh::u128)1) << 64) / q);
  pd.mu63 = (uint64_t)((((hxh::u128)1) << (63 + pd.k)) / q);
  pd.tw_fwd_off = ph.tw_fwd_off;
  pd.tw_inv_off = ph.tw_inv_off;
  pd.proth = ph.proth ? 1u : 0u;
  pd.r2 = hx::tw_mont_form(hx::tw_mont_form(1 % q, q), q);
  int idx = (int)c->primes.size();
  HIPCHK(hipMemcpy(c->d_primes + idx, &pd, sizeof pd, hipMemcpyHostToDevice));
  c->primes.push_back(ph);
  if (c->pow2)
    c->bigplan.push_back(big);
  if (!c->pow2) {
    int rc = blue_prime_create(c, idx);
    if (rc != HX_OK) {
      c->primes.pop_back();
      return rc;
    }
  }
  if (idx_out)
    *idx_out = idx;
  return HX_OK;
}

// ------------------------------------------------------------------
// polys
// ------------------------------------------------------------------
static int check_rows(hx_ctx* c, const int* idx, int n, bool allow_dup = false)
{
  if (n < 0 || (n > 0 && !idx))
    return fail(HX_ERR_INVALID, "bad prime index list");
  for (int i = 0; i < n; i++) {
    if (idx[i] < 0 || idx[i] >= (int)c->primes.size())
      return fail(HX_ERR_INVALID, "prime index %d not in the context", idx[i]);
    if (!allow_dup)
      for (int j = 0; j < i; j++)
        if (idx[j] == idx[i])
          return fail(HX_ERR_INVALID, "duplicate prime index %d", idx[i]);
  }
  return HX_OK;
}

static int poly_new(hx_ctx* c, int batch, const int* idx, int nrows, int cap, void* wrap,
                    bool allow_dup, hx_poly** out, bool zero = true)
{
  if (!c || !out || batch < 1)
    return fail(HX_ERR_INVALID, "bad argument");
  CTX_ENTER(c);   // (before check_rows: it reads the prime list a concurrent hx_ctx_add_prime grows)
  CHK(check_rows(c, idx, nrows, allow_dup));
  if (cap < nrows)
    cap = nrows;
  if (cap < 1)
    cap = 1;
  if (!wrap)
    cap += 2;  // room for a mod-up by a couple of primes without reallocating
  hx_poly* p = new hx_poly();
  p->ctx = c;
  p->batch = batch;
  p->cap_rows = cap;
  p->prime_idx.assign(idx, idx + nrows);
  p->owns = wrap == nullptr;
  if (wrap) {
    p->d = (uint64_t*)wrap;
  } else {
    size_t bytes = (size_t)cap * batch * c->phim * 8;
    hipError_t e = pool_alloc(c, bytes, (void**)&p->d);
    if (e != hipSuccess) {
      delete p;
      return fail(HX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    // DoubleCRT(context, set) is zero-initialised: only the live rows need it
    e = zero ? hipMemsetAsync(p->d, 0, (size_t)nrows * batch * c->phim * 8, c->stream) : hipSuccess;
    if (e != hipSuccess) {
      pool_free(c, p->d, bytes);
      delete p;
      return fail(HX_ERR_DEVICE, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
  }
  c->refs++;
  c->polys.insert(p);
  *out = p;
  return HX_OK;
}

extern "C" int hx_poly_create(hx_ctx* c, int batch, const int* idx, int nrows, hx_poly** out)
{
  return poly_new(c, batch, idx, nrows, nrows, nullptr, false, out);
}
extern "C" int hx_poly_create_uninit(hx_ctx* c, int batch, const int* idx, int nrows, hx_poly** out)
{
  return poly_new(c, batch, idx, nrows, nrows, nullptr, false, out, false);
}
extern "C" int hx_poly_wrap(hx_ctx* c, int batch, const int* idx, int nrows, void* dptr,
                            hx_poly** out)
{
  if (!dptr)
    return fail(HX_ERR_INVALID, "null device pointer");
  return poly_new(c, batch, idx, nrows, nrows, dptr, false, out);
}
extern "C" int hx_poly_destroy(hx_poly* p)
{
  if (!p)
    return HX_OK;
  {
    std::lock_guard<std::recursive_mutex> lk(p->ctx->mu);
    hipSetDevice(p->ctx->device);
    storage_release(p);
    p->ctx->polys.erase(p);
  }
  ctx_release(p->ctx);
  delete p;
  return HX_OK;
}
extern "C" int hx_poly_shape(const hx_poly* p, int* batch, int* nrows, uint64_t* phim)
{
  if (!p)
    return fail(HX_ERR_INVALID, "null poly");
  if (batch)
    *batch = p->batch;
  if (nrows)
    *nrows = p->nrows();
  if (phim)
    *phim = p->ctx->phim;
  return HX_OK;
}
extern "C" int hx_poly_primes(const hx_poly* p, int* out)
{
  if (!p || !out)
    return fail(HX_ERR_INVALID, "null argument");
  for (int i = 0; i < p->nrows(); i++)
    out[i] = p->prime_idx[i];
  return HX_OK;
}
extern "C" void* hx_poly_device_ptr(hx_poly* p)
{
  if (!p)
    return nullptr;
  std::lock_guard<std::recursive_mutex> lk(p->ctx->mu);
  hipSetDevice(p->ctx->device);
  if (poly_own(p) != HX_OK)  // the caller may write through the pointer
    return nullptr;
  p->exposed = true;
  return p->d;
}

extern "C" int hx_poly_upload(hx_poly* p, const uint64_t* host)
{
  if (!p || !host)
    return fail(HX_ERR_INVALID, "null argument");
  CTX_ENTER(p->ctx);
  NO_CAPTURE(p->ctx, "hx_poly_upload");
  OWN(p);
  size_t bytes = (size_t)p->nrows() * p->row_words() * 8;
  HIPCHK(hipMemcpyAsync(p->d, host, bytes, hipMemcpyHostToDevice, p->ctx->stream));
  HIPCHK(hipStreamSynchronize(p->ctx->stream));
  return HX_OK;
}
extern "C" int hx_poly_download(const hx_poly* p, uint64_t* host)
{
  if (!p || !host)
    return fail(HX_ERR_INVALID, "null argument");
  CTX_ENTER(p->ctx);
  NO_CAPTURE(p->ctx, "hx_poly_download");
  size_t bytes = (size_t)p->nrows() * p->row_words() * 8;
  HIPCHK(hipMemcpyAsync(host, p->d, bytes, hipMemcpyDeviceToHost, p->ctx->stream));
  HIPCHK(hipStreamSynchronize(p->ctx->stream));
  return HX_OK;
}

// keep = false: the caller overwrites every row (pure output), so the old contents are not copied
static int poly_reserve(hx_poly* p, int cap, bool keep = true)
{
  if (cap <= p->cap_rows)
    return HX_OK;
  if (!p->owns)
    return fail(HX_ERR_NOMEM, "wrapped poly has no room for %d rows", cap);
  hx_ctx* c = p->ctx;
  uint64_t* nd = nullptr;
  cap += 2;
  size_t bytes = (size_t)cap * p->row_words() * 8;
  hipError_t e = pool_alloc(c, bytes, (void**)&nd);
  if (e != hipSuccess)
    return fail(HX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  if (keep && p->nrows() > 0)
    CHK(dcopy(c, nd, p->d, (size_t)p->nrows() * p->row_words()));
  storage_release(p);  // stream-ordered reuse (a shared slab stays with its other holders)
  p->d = nd;
  p->cap_rows = cap;
  return HX_OK;
}

extern "C" int hx_poly_copy(hx_poly* dst, const hx_poly* src)
{
  if (!dst || !src || dst->ctx != src->ctx || dst->batch != src->batch)
    return fail(HX_ERR_INVALID, "Context mismatch");
  CTX_ENTER(dst->ctx);
  if (dst == src)
    return HX_OK;
  // Lazy unless a HIP graph is involved or a raw pointer is out: a graph replays on the addresses it
  // recorded ("inputs are whatever the input polys hold at replay time"), which only holds if an
  // input poly never has to move off its slab to be written -- so nothing is shared while a capture
  // is open or a graph is alive (hx_ctx_graph_begin also un-shares what was shared before).
  hx_ctx* cc = dst->ctx;
  const bool lazy_ok = !cc->capturing && cc->graphs_alive == 0 && !dst->exposed && !src->exposed;
  if (dst->owns && src->owns && lazy_ok) {
    // lazy: share the source's slab (see hx_poly::Share); nothing moves until somebody writes
    hx_poly* s = const_cast<hx_poly*>(src);
    if (dst->d == s->d)
      return HX_OK;  // already sharing
    storage_release(dst);
    if (!s->share)
      s->share = new hx_poly::Share{1};
    s->share->refs++;
    dst->share = s->share;
    dst->d = s->d;
    dst->cap_rows = s->cap_rows;
    dst->prime_idx = s->prime_idx;
    return HX_OK;
  }
  OWN(dst);
  CHK(poly_reserve(dst, src->nrows(), /*keep=*/false));
  dst->prime_idx = src->prime_idx;
  CHK(dcopy(dst->ctx, dst->d, src->d, (size_t)src->nrows() * src->row_words()));
  return HX_OK;
}
extern "C" int hx_poly_set_zero(hx_poly* p)
{
  if (!p)
    return fail(HX_ERR_INVALID, "null poly");
  CTX_ENTER(p->ctx);
  OWN(p);
  HIPCHK(hipMemsetAsync(p->d, 0, (size_t)p->nrows() * p->row_words() * 8, p->ctx->stream));
  return HX_OK;
}

// DoubleCRT::randomize (src/DoubleCRT.cpp:1258-1378) on the device: prg_kernels.h
extern "C" int hx_randomize(hx_poly* p, const uint8_t* key32, uint64_t stream)
{
  if (!p || !key32)
    return fail(HX_ERR_INVALID, "null argument");
  hx_ctx* c = p->ctx;
  CTX_ENTER(c);
  if (p->nrows() == 0)
    return HX_OK;
  if (p->nrows() > MAX_ROWS || p->batch > 65535)
    return fail(HX_ERR_UNSUPPORTED, "hx_randomize: more than %d rows or 65535 batch elements", MAX_ROWS);
  OWN(p);
  hx::RandArgs A;
  memset(&A, 0, sizeof A);
  A.data = p->d;
  for (int i = 0; i < 8; i++)
    A.key[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) | ((uint32_t)key32[4 * i + 2] << 16) |
               ((uint32_t)key32[4 * i + 3] << 24);
  A.stream_lo = (uint32_t)stream;
  A.stream_hi = (uint32_t)(stream >> 32);
  A.phim = c->phim;
  A.batch = p->batch;
  for (int r = 0; r < p->nrows(); r++) {
    if (p->prime_idx[r] > 65535)
      return fail(HX_ERR_UNSUPPORTED, "hx_randomize: prime index above 65535");
    A.rows.p[r] = (uint16_t)p->prime_idx[r];
  }
  HX_LAUNCH(hx::randomize_kernel, dim3((unsigned)p->nrows() * (unsigned)p->batch), dim3(256), 0,
                     c->stream, A, c->d_primes);
  HIPCHK(hipGetLastError());
  return HX_OK;
}

static int find_row(const std::vector<int>& v, int prime)
{
  for (size_t i = 0; i < v.size(); i++)
    if (v[i] == prime)
      return (int)i;
  return -1;
}

extern "C" int hx_poly_remove_primes(hx_poly* p, const int* idx, int n)
{
  if (!p || (n > 0 && !idx))
    return fail(HX_ERR_INVALID, "null argument");
  CTX_ENTER(p->ctx);
  OWN(p);
  std::vector<int> keep;
  size_t rw = p->row_words();
  int w = 0;
  for (int r = 0; r < p->nrows(); r++) {
    bool drop = false;
    for (int i = 0; i < n; i++)
      if (idx[i] == p->prime_idx[r])
        drop = true;
    if (drop)
      continue;
    if (w != r)
      CHK(dcopy(p->ctx, p->d + (size_t)w * rw, p->d + (size_t)r * rw, rw));
    keep.push_back(p->prime_idx[r]);
    w++;
  }
  p->prime_idx = keep;
  return HX_OK;
}

// ------------------------------------------------------------------
// NTT
// ------------------------------------------------------------------
static int make_map(const std::vector<int>& primes, int first, int count, RowMap& map)
{
  if (count > MAX_ROWS)
    return fail(HX_ERR_UNSUPPORTED, "more than %d rows per launch descriptor", MAX_ROWS);
  for (int r = 0; r < count; r++)
    map.p[r] = (uint16_t)primes[first + r];
  return HX_OK;
}

// raw launch: listed (row, table-entry) pairs of a [rows][batch][2^logn] buffer, in -> out
// lazy_in (forward, N = 2^13..2^15 only): the rows hold words in [0,8q), e.g. the unreduced output of
// break_digits_fast_kernel<., true>
static int ntt_launch(hx_ctx* c, int logn, const PrimeDev* table, const uint64_t* in, uint64_t* out,
                      const std::vector<std::pair<int, int>>& rows, int batch, bool inverse, bool lazy_in)
{
  if (rows.empty())
    return HX_OK;
  if (logn < 1 || logn > 15)
    return fail(HX_ERR_UNSUPPORTED, "negacyclic NTT kernels support sizes 2..32768");
  // N = 2^15 forward, out of place, on the context's own primes: two 2^14-point workgroups per row
  bool half15 = logn == 15 && !inverse && table == c->d_primes && in != out && c->sw.half15;
  for (size_t i = 0; i < rows.size() && half15; i++)
    half15 = rows[i].second >= 0 && rows[i].second < (int)c->primes.size() && c->primes[(size_t)rows[i].second].half_pd >= 0 &&
             c->primes[(size_t)rows[i].second].half_pd < 0xffff;
  for (size_t first = 0; first < rows.size(); first += MAX_ROWS) {
    int n = (int)std::min<size_t>(MAX_ROWS, rows.size() - first);
    NttRows d;
    for (int i = 0; i < n; i++) {
      if (rows[first + i].first > 0xffff || rows[first + i].second > 0xffff)
        return fail(HX_ERR_UNSUPPORTED, "row index too large for one launch descriptor");
      d.row[i] = (uint16_t)rows[first + i].first;
      d.prime[i] = half15 ? (uint16_t)c->primes[(size_t)rows[first + i].second].half_pd : (uint16_t)rows[first + i].second;
    }
    if (half15) {
      hipError_t e = hx::launch_ntt_half15_fwd(lazy_in, in, out, d, n, batch, c->d_cprimes, c->d_tw, c->stream);
      if (e != hipSuccess)
        return fail(HX_ERR_DEVICE, "NTT launch failed: %s", hipGetErrorString(e));
      continue;
    }
    hipError_t e = (lazy_in && !inverse && logn >= 13 && logn <= 15)
                       ? hx::launch_ntt_pow2_lazy_in(logn, in, out, d, n, batch, table, c->d_tw, c->stream)
                       : hx::launch_ntt_pow2(logn, inverse, in, out, d, n, batch, table, c->d_tw, c->stream);
    if (e != hipSuccess)
      return fail(HX_ERR_DEVICE, "NTT launch failed: %s", hipGetErrorString(e));
  }
  return HX_OK;
}

static int bluestein_rows(hx_ctx* c, const uint64_t* in, uint64_t* out,
                          const std::vector<std::pair<int, int>>& rows, int batch, bool inverse);

// power-of-two rings with N = 2^16..2^19: split into 4 / 8 / 16 sub-transforms on the row kernels, natural
// order in and out (kernels and layout: bluestein.h big_pre / big_post); in == out is allowed
static int pow2_big_rows(hx_ctx* c, const uint64_t* in, uint64_t* out,
                         const std::vector<std::pair<int, int>>& rows, int batch, bool inverse)
{
  const int logn = c->logn;
  const int S = logn > 18 ? 16 : (logn > 17 ? 8 : 4), lsub = logn - (S == 16 ? 4 : (S == 8 ? 3 : 2));
  const uint32_t Q = 1u << lsub;
  const size_t row_words = (size_t)batch << logn;
  // rows per pass: one launch descriptor, and at most 512 MiB of sub-block scratch
  size_t chunk = std::max<size_t>(1, std::min<size_t>(MAX_ROWS / S, ((size_t)64 << 20) / row_words));
  CHK(ensure_scratch(c, 9, std::min(chunk, rows.size()) * row_words));
  uint64_t* qbuf = c->scratch[9];
  for (size_t first = 0; first < rows.size(); first += chunk) {
    const int R = (int)std::min(chunk, rows.size() - first);
    NttRows d;
    hx::PtrList cps;
    std::vector<std::pair<int, int>> sub;
    for (int r = 0; r < R; r++) {
      const auto& rp = rows[first + r];
      if (rp.first > 0xffff)
        return fail(HX_ERR_UNSUPPORTED, "row index too large for one launch descriptor");
      d.row[r] = (uint16_t)rp.first;
      d.prime[r] = (uint16_t)rp.second;
      const ConvPlan& pl = c->bigplan[(size_t)rp.second];
      cps.p[r] = pl.dev;
      for (int g = 0; g < S; g++)
        sub.emplace_back(r * S + g, pl.pd[g]);
    }
    const dim3 grid = grid2(Q, (size_t)R * batch);
    if (S == 16)
      HX_LAUNCH(hx::big_pre_kernel<16>, grid, dim3(256), 0, c->stream, in, qbuf, d, cps, batch, Q, inverse ? 1 : 0);
    else if (S == 8)
      HX_LAUNCH(hx::big_pre_kernel<8>, grid, dim3(256), 0, c->stream, in, qbuf, d, cps, batch, Q, inverse ? 1 : 0);
    else
      HX_LAUNCH(hx::big_pre_kernel<4>, grid, dim3(256), 0, c->stream, in, qbuf, d, cps, batch, Q, inverse ? 1 : 0);
    HIPCHK(hipGetLastError());
    CHK(ntt_launch(c, lsub, c->d_cprimes, qbuf, qbuf, sub, batch, inverse));
    if (S == 16)
      HX_LAUNCH(hx::big_post_kernel<16>, grid, dim3(256), 0, c->stream, qbuf, out, d, cps, batch, Q, inverse ? 1 : 0);
    else if (S == 8)
      HX_LAUNCH(hx::big_post_kernel<8>, grid, dim3(256), 0, c->stream, qbuf, out, d, cps, batch, Q, inverse ? 1 : 0);
    else
      HX_LAUNCH(hx::big_post_kernel<4>, grid, dim3(256), 0, c->stream, qbuf, out, d, cps, batch, Q, inverse ? 1 : 0);
    HIPCHK(hipGetLastError());
  }
  return HX_OK;
}

// Cmodulus::FFT / iFFT on the listed (row, prime) pairs of a [rows][batch][phi(m)] buffer
// a ring whose forward row transform takes lazy input (ntt_launch lazy_in): the exact-RNS kernels in front of it may
// leave their output words unreduced
static bool ntt_lazy_input_ok(const hx_ctx* c) { return c->pow2 && c->logn >= 13 && c->logn <= 15; }
// the digit kernel may leave its extension words unreduced for that transform: rows of Proth-form primes are then read
// at bound 2 (ntt_kernels.hip BufIOT), which is what the digit kernel's Proth-form target sums deliver -- with those
// switched off (HX_NO_PROTH_RNS) while the rows keep the Proth-form butterflies, the words are reduced instead
static bool digits_lazy_ok(const hx_ctx* c) { return ntt_lazy_input_ok(c) && (!c->sw.no_proth_rns || c->sw.no_proth); }

static int ntt_list(hx_ctx* c, const uint64_t* in, uint64_t* out,
                    const std::vector<std::pair<int, int>>& rows, int batch, bool inverse, bool lazy_in = false)
{
  if (rows.empty())
    return HX_OK;
  if (!c->pow2)
    return bluestein_rows(c, in, out, rows, batch, inverse);
  if (c->logn >= 16 && c->logn <= 19)
    return pow2_big_rows(c, in, out, rows, batch, inverse);
  if (c->logn < 1 || c->logn > 19)
    return fail(HX_ERR_UNSUPPORTED, "power-of-two NTT supports 2 <= phi(m) <= 524288");
  return ntt_launch(c, c->logn, c->d_primes, in, out, rows, batch, inverse, lazy_in && ntt_lazy_input_ok(c));
}

// in place: rows [row0,row0+nrows); row r uses prime plist[r % period]
static int ntt_rows(hx_ctx* c, uint64_t* data, const std::vector<int>& plist, int period,
                    int row0, int nrows, int batch, bool inverse)
{
  std::vector<std::pair<int, int>> rows;
  rows.reserve(nrows);
  for (int r = row0; r < row0 + nrows; r++)
    rows.emplace_back(r, plist[r % period]);
  return ntt_list(c, data, data, rows, batch, inverse);
}

// all rows of p; a shared p (copy-on-write) is transformed out of place into its own new slab
static int ntt_poly(hx_poly* p, bool inverse)
{
  const uint64_t* src;
  CHK(poly_fresh(p, &src));
  std::vector<std::pair<int, int>> rows;
  rows.reserve(p->nrows());
  for (int r = 0; r < p->nrows(); r++)
    rows.emplace_back(r, p->prime_idx[r]);
  int rc = ntt_list(p->ctx, src, p->d, rows, p->batch, inverse);
  if (rc != HX_OK && src != p->d)  // the poly left a shared slab for a fresh one: it keeps its old rows
    (void)dcopy(p->ctx, p->d, src, (size_t)p->nrows() * p->row_words());
  return rc;
}

extern "C" int hx_ntt_forward(hx_poly* p)
{
  if (!p)
    return fail(HX_ERR_INVALID, "null poly");
  CTX_ENTER(p->ctx);
  return ntt_poly(p, false);
}
extern "C" int hx_ntt_inverse(hx_poly* p)
{
  if (!p)
    return fail(HX_ERR_INVALID, "null poly");
  CTX_ENTER(p->ctx);
  return ntt_poly(p, true);
}

extern "C" int hx_time_ntt(hx_poly* p, int dir, int iters, int max_rows, float* avg_ms)
{
  if (!p || !avg_ms || iters < 1)
    return fail(HX_ERR_INVALID, "bad argument");
  hx_ctx* c = p->ctx;
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_time_ntt");
  OWN(p);
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, c->stream));
  for (int i = 0; i < iters; i++) {
    int nr = (max_rows > 0 && max_rows < p->nrows()) ? max_rows : p->nrows();
    int rc = ntt_rows(c, p->d, p->prime_idx, p->nrows(), 0, nr, p->batch, dir != 0);
    if (rc != HX_OK)
      return rc;
  }
  HIPCHK(hipEventRecord(e1, c->stream));
  HIPCHK(hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, e0, e1));
  *avg_ms = ms / iters;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return HX_OK;
}

// HIP events on the context's own stream around whatever the caller enqueues in between (the
// bench times a launch set such as one hx_bring_to_set_multi this way; torch.cuda.Event would see
// torch's current stream only).
extern "C" int hx_ctx_timer_begin(hx_ctx* c)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null context");
  CTX_ENTER(c);
  if (!c->timer[0]) {
    HIPCHK(hipEventCreate(&c->timer[0]));
    HIPCHK(hipEventCreate(&c->timer[1]));
  }
  HIPCHK(hipEventRecord(c->timer[0], c->stream));
  return HX_OK;
}
extern "C" int hx_ctx_timer_end(hx_ctx* c, float* ms)
{
  if (!c || !ms)
    return fail(HX_ERR_INVALID, "bad argument");
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_ctx_timer_end");
  if (!c->timer[0])
    return fail(HX_ERR_INVALID, "hx_ctx_timer_end without hx_ctx_timer_begin");
  HIPCHK(hipEventRecord(c->timer[1], c->stream));
  HIPCHK(hipEventSynchronize(c->timer[1]));
  HIPCHK(hipEventElapsedTime(ms, c->timer[0], c->timer[1]));
  return HX_OK;
}

// In-situ kernel timing (prof.h): between begin and end every kernel the library launches, on any
// context of this process, is bracketed by HIP events on its own stream.  hx_profile_end waits for
// them and writes a JSON summary (per kernel and launch size: calls, total / avg / min / max
// microseconds) into `json`; when `cap` is too small nothing is written and *needed tells the size.
extern "C" int hx_profile_begin(void)
{
  hxp::begin();
  return HX_OK;
}
extern "C" int hx_profile_end(char* json, size_t cap, size_t* needed)
{
  // a size query or a too-small buffer leaves the summary in place (hxp::State, under its mutex: any thread may
  // fetch it; hx_profile_begin drops an unfetched one); recording stops at the first call either way
  const std::string sum = hxp::end(/*fetch=*/false);
  if (needed)
    *needed = sum.size() + 1;
  if (!json || cap < sum.size() + 1)
    return json ? fail(HX_ERR_INVALID, "hx_profile_end: buffer of %zu bytes, %zu needed", cap, sum.size() + 1) : HX_OK;
  memcpy(json, sum.c_str(), sum.size() + 1);
  (void)hxp::end(/*fetch=*/true);
  return HX_OK;
}

// ------------------------------------------------------------------
// HIP graphs: a sequence of engine calls captured once and replayed with one launch -- for the
// launch-bound case (one ciphertext at a time, as benchmarks/bgv_basic.cpp:158-164 runs: ~40 kernels
// of a few microseconds each per multiply).  Everything enqueued on the context between begin and
// end is recorded instead of run.  A replay re-executes exactly those kernels on exactly those
// buffers: the inputs are whatever the input polys hold at replay time, the outputs land in the polys
// the captured calls returned.  While a graph is alive the context keeps every buffer the graph may
// point at (pinned arena blocks / hx_ctx::graph_retired).  Calls that must wait for the device
// (downloads, uploads, norm read-backs) cannot be captured: HIP fails them, and so does end().
// ------------------------------------------------------------------
struct hx_graph {
  hx_ctx* ctx;
  hipGraph_t graph;
  hipGraphExec_t exec;
};
// no recording open and no graph alive: nothing points at the kept buffers any more -- slabs go back
// to the pool, replaced buffers are freed (the stream is drained first)
static void graph_release_if_idle(hx_ctx* c)
{
  if (c->capturing || c->graphs_alive > 0)
    return;
  const bool parked = !c->arena.deferred.empty() || !c->graph_retired.empty();
  if (parked)
    hipStreamSynchronize(c->stream);
  c->arena.unpin_all();
  for (void* q : c->graph_retired)
    hipFree(q);
  c->graph_retired.clear();
}
extern "C" int hx_ctx_graph_begin(hx_ctx* c)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null context");
  CTX_ENTER(c);
  if (c->capturing)
    return fail(HX_ERR_INVALID, "a graph is already being captured on this context");
  if (!c->norm_pending.empty())
    return fail(HX_ERR_INVALID, "deferred norms are pending: hx_norms_flush before capturing");
  if (!c->stream) {
    // the legacy default stream cannot be captured: from here on the context runs on a stream of
    // its own (everything enqueued so far is complete after the device-wide wait)
    HIPCHK(hipDeviceSynchronize());
    if (!c->own_stream)
      HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
  }
  // every poly gets a slab of its own before the recording starts: a recorded kernel that reads an
  // input poly must keep seeing that poly's data at replay time, which a copy-on-write move of the
  // poly to another slab (at its next upload / set_zero / element-wise write) would break
  for (hx_poly* p : c->polys)
    if (p->share)
      CHK(poly_own(p));
  // whatever is live now may be referenced by the recording
  c->arena.pin_all();
  HIPCHK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
  c->capturing = true;
  return HX_OK;
}
extern "C" int hx_ctx_graph_end(hx_ctx* c, hx_graph** out)
{
  if (!c || !out)
    return fail(HX_ERR_INVALID, "null argument");
  CTX_ENTER(c);
  if (!c->capturing)
    return fail(HX_ERR_INVALID, "hx_ctx_graph_end without hx_ctx_graph_begin");
  c->capturing = false;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(c->stream, &g);
  if (e != hipSuccess || !g) {
    (void)hipGetLastError();
    graph_release_if_idle(c);
    return fail(HX_ERR_DEVICE, "graph capture failed (a captured call needed the device to finish?): %s",
                hipGetErrorString(e));
  }
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    hipGraphDestroy(g);
    graph_release_if_idle(c);
    return fail(HX_ERR_DEVICE, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  }
  hx_graph* h = new hx_graph();
  h->ctx = c;
  h->graph = g;
  h->exec = x;
  c->graphs_alive++;
  c->refs++;
  *out = h;
  return HX_OK;
}
extern "C" int hx_graph_launch(hx_graph* g)
{
  if (!g)
    return fail(HX_ERR_INVALID, "null graph");
  hx_ctx* c = g->ctx;
  CTX_ENTER(c);
  if (c->capturing)
    return fail(HX_ERR_INVALID, "hx_graph_launch while another graph is being captured");
  HIPCHK(hipGraphLaunch(g->exec, c->stream));
  return HX_OK;
}
extern "C" int hx_graph_destroy(hx_graph* g)
{
  if (!g)
    return HX_OK;
  hx_ctx* c = g->ctx;
  {
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    hipGraphExecDestroy(g->exec);
    hipGraphDestroy(g->graph);
    --c->graphs_alive;
    graph_release_if_idle(c);
  }
  ctx_release(c);
  delete g;
  return HX_OK;
}

// ------------------------------------------------------------------
// element-wise
// ------------------------------------------------------------------
static dim3 ew_grid(size_t row_words, int rows)
{
  size_t nvec = row_words / 2;
  size_t blocks = (nvec + 255) / 256;
  size_t cap = 8192 / (size_t)(rows > 0 ? rows : 1) + 1;  // ~ 32 blocks per CU overall
  if (blocks > cap)
    blocks = cap;
  if (blocks < 1)
    blocks = 1;
  return dim3((unsigned)blocks, (unsigned)rows);
}

template <int OP>
static int ew_binary(hx_poly* a, const hx_poly* b)
{
  if (!a || !b)
    return fail(HX_ERR_INVALID, "null poly");
  if (a->ctx != b->ctx)
    return fail(HX_ERR_INVALID, "DoubleCRT::Op: incompatible objects");
  if (b->batch != a->batch && b->batch != 1)
    return fail(HX_ERR_INVALID, "batch mismatch");
  CTX_ENTER(a->ctx);
  int rows = a->nrows();
  if (rows == 0)
    return HX_OK;
  if (rows > MAX_ROWS)
    return fail(HX_ERR_UNSUPPORTED, "too many rows");
  RowMap2 map;
  for (int r = 0; r < rows; r++) {
    int br = find_row(b->prime_idx, a->prime_idx[r]);
    if (br < 0)
      return fail(HX_ERR_PRIMESET, "DoubleCRT::Op: incompatible index sets (prime %d)",
                  a->prime_idx[r]);
    map.p[r] = (uint16_t)a->prime_idx[r];
    map.brow[r] = (uint16_t)br;
  }
  size_t rw = a->row_words();
  OWN(a);
  HX_LAUNCH((hx::ew_binary_kernel<OP>), ew_grid(rw, rows), dim3(256), 0, a->ctx->stream,
                     a->d, b->d, map, rw, b->row_words(), (int)(b->batch != a->batch),
                     (size_t)a->ctx->phim, a->ctx->d_primes);
  HIPCHK(hipGetLastError());
  return HX_OK;
}
extern "C" int hx_add(hx_poly* a, const hx_poly* b) { return ew_binary<hx::EW_ADD>(a, b); }
extern "C" int hx_sub(hx_poly* a, const hx_poly* b) { return ew_binary<hx::EW_SUB>(a, b); }
extern "C" int hx_mul(hx_poly* a, const hx_poly* b) { return ew_binary<hx::EW_MUL>(a, b); }

template <int OP>
static int ew_scalar_rows(hx_poly* a, const uint64_t* c_per_row, uint64_t exponent = 0)
{
  if (!a)
    return fail(HX_ERR_INVALID, "null poly");
  CTX_ENTER(a->ctx);
  int rows = a->nrows();
  if (rows == 0)
    return HX_OK;
  if (rows > MAX_ROWS)
    return fail(HX_ERR_UNSUPPORTED, "too many rows");
  RowMap map;
  RowScalars sc;
  for (int r = 0; r < rows; r++) {
    uint64_t q = a->ctx->primes[a->prime_idx[r]].q;
    map.p[r] = (uint16_t)a->prime_idx[r];
    uint64_t cv = c_per_row ? c_per_row[r] % q : 0;
    sc.c[r] = cv;
    sc.cp[r] = OP == hx::EWS_EXP ? exponent : hxh::shoup(cv, q);
  }
  size_t rw = a->row_words();
  OWN(a);
  HX_LAUNCH((hx::ew_scalar_kernel<OP>), ew_grid(rw, rows), dim3(256), 0, a->ctx->stream,
                     a->d, map, sc, rw, a->ctx->d_primes);
  HIPCHK(hipGetLastError());
  return HX_OK;
}
extern "C" int hx_add_scalar(hx_poly* a, const uint64_t* c)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null scalars");
  return ew_scalar_rows<hx::EWS_ADD>(a, c);
}
extern "C" int hx_sub_scalar(hx_poly* a, const uint64_t* c)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null scalars");
  return ew_scalar_rows<hx::EWS_SUB>(a, c);
}
extern "C" int hx_mul_scalar(hx_poly* a, const uint64_t* c)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null scalars");
  return ew_scalar_rows<hx::EWS_MUL>(a, c);
}
extern "C" int hx_negate(hx_poly* a) { return ew_scalar_rows<hx::EWS_NEG>(a, nullptr); }
extern "C" int hx_set_scalar(hx_poly* a, const uint64_t* c)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null scalars");
  return ew_scalar_rows<hx::EWS_SET>(a, c);
}
extern "C" int hx_exp(hx_poly* a, uint64_t e)
{
  if (!a)
    return fail(HX_ERR_INVALID, "null poly");
  std::vector<uint64_t> zero((size_t)std::max(1, a->nrows()), 0);
  return ew_scalar_rows<hx::EWS_EXP>(a, zero.data(), e);
}

extern "C" int hx_automorph(hx_poly* a, uint64_t k)
{
  if (!a)
    return fail(HX_ERR_INVALID, "null poly");
  hx_ctx* c = a->ctx;
  CTX_ENTER(c);
  k %= c->m;
  if (hxh::gcd(k, c->m) != 1)
    return fail(HX_ERR_NOT_IN_ZMSTAR, "DoubleCRT::automorph: k not in Zm*");
  int rows = a->nrows();
  if (rows == 0)
    return HX_OK;
  size_t words = (size_t)rows * a->row_words();
  // a shared poly (copy-on-write) is permuted straight from the shared slab into its own new one;
  // otherwise through a scratch copy
  const uint64_t* src;
  CHK(poly_fresh(a, &src));
  const bool direct = src != a->d;
  if (!direct)
    CHK(ensure_scratch(c, 3, words));
  if (!c->pow2) {
    HX_LAUNCH(hx::perm_build_kernel, dim3((c->phim + 255) / 256), dim3(256), 0, c->stream,
                       c->d_perm, c->d_zms, c->d_zms_index, c->phim, c->m, k);
    HIPCHK(hipGetLastError());
  }
  size_t nseg = (size_t)rows * a->batch;
  unsigned bx = (c->phim + 255) / 256;
  if (bx > 64)
    bx = 64;
  HX_LAUNCH(hx::gather_kernel, dim3(bx, (unsigned)nseg), dim3(256), 0, c->stream,
                     direct ? a->d : c->scratch[3], src, c->d_perm, c->phim, nseg, (int)c->pow2, c->m, k);
  HIPCHK(hipGetLastError());
  if (!direct)
    CHK(dcopy(c, a->d, c->scratch[3], words));
  return HX_OK;
}
extern "C" int hx_complex_conj(hx_poly* a)
{
  if (!a)
    return fail(HX_ERR_INVALID, "null poly");
  return hx_automorph(a, a->ctx->m - 1);  // src/DoubleCRT.cpp:1239
}

// ------------------------------------------------------------------
// exact RNS plans
// ------------------------------------------------------------------
// tgt_moduli (optional): explicit target moduli instead of context primes (tgt then only sizes the
// plan): any t in [2, 2^60) -- a target needs no transform tables (hx_poly_rem).
// scaled: every target residue comes out multiplied by P^-1 mod q_t (W[t][k] and P mod t are
// stored times P^-1: the kernels themselves are unchanged) -- the several-primes mod-switch wants
// delta / P, whose transform it subtracts from c_r / P.
static int get_plan(hx_ctx* c, const std::vector<int>& src, const std::vector<int>& tgt,
                    uint64_t ptxt, ExtPlan** out, const std::vector<uint64_t>* tgt_moduli = nullptr,
                    bool scaled = false)
{
  int n = (int)src.size(), nt = (int)tgt.size();
  if (n < 1 || n > MAX_EXT_SRC)
    return fail(HX_ERR_UNSUPPORTED, "basis extension supports 1..%d source primes (got %d)", MAX_EXT_SRC, n);
  if (nt > MAX_ROWS)
    return fail(HX_ERR_UNSUPPORTED, "too many target primes");
  std::vector<uint64_t> key;
  key.push_back((uint64_t)n);
  for (int s : src)
    key.push_back((uint64_t)s);
  for (int t : tgt)
    key.push_back((uint64_t)t);
  key.push_back(ptxt);
  if (scaled)
    key.push_back(0x5ca1edull << 40);
  if (tgt_moduli) {
    key.push_back(0x7e57ull << 48);
    for (uint64_t t : *tgt_moduli)
      key.push_back(t);
  }
  auto tq = [&](int t) { return tgt_moduli ? (*tgt_moduli)[t] : c->primes[tgt[t]].q; };
  auto it = c->plans.find(key);
  if (it != c->plans.end()) {
    *out = it->second;
    return HX_OK;
  }
  std::vector<uint64_t> p(n);
  for (int k = 0; k < n; k++)
    p[k] = c->primes[src[k]].q;
  // blob layout in 64-bit words
  size_t off = 0;
  auto take = [&](size_t words) {
    size_t o = off;
    off += (words + 1) & ~(size_t)1;  // keep 16-byte alignment
    return o;
  };
  size_t o_srcq = take(n), o_srcmu = take(n), o_ginv = take((size_t)2 * n * n), o_half = take(n);
  size_t o_tq = take(nt), o_tmu64 = take(nt), o_tmu = take(nt), o_tk = take((nt + 1) / 2);
  size_t o_pmod = take(nt), o_W = take((size_t)2 * nt * n), o_upd = take((size_t)2 * nt);
  size_t o_Wp = take((size_t)2 * n);
  size_t o_tlazy = take((nt + 1) / 2);
  size_t o_srcrq = take(n), o_tmu63 = take(nt);
  size_t o_tchunk = take((nt + 1) / 2);
  const size_t pack_stride = 10 + 2 * (size_t)n;  // per-target record of the fast kernels (TgtRec in rns_kernels.h)
  size_t o_pack = take((size_t)nt * pack_stride);
  size_t o_pack2 = take((size_t)nt * pack_stride);   // the same for the HPS front end (rns_kernels.h)
  size_t o_hinv = take((size_t)2 * n), o_Wp2 = take((size_t)2 * n);
  size_t o_ginvm = take((size_t)n * n);   // Garner constants times 2^64 (Proth-form sources, ExtPlanDev::ginv_m)
  // rns_extend_wide_kernel (17..40 source primes): per-target record of 8 + n words, padded so that the last
  // record's group-of-four multiplier reads stay inside the blob
  const bool wide_cand = n > 16 && n <= 40 && !c->sw.no_wide_extend;
  // 9 .. 16 sources (the dropped primes of a CKKS level-2 mod-switch): the fast kernels' plans, whose HPS-form launches
  // go to the matrix-core kernel too -- it reads the wide record's header for its rarer constants
  const bool mfma_small = n >= c->sw.mfma_min_n && n <= 16 && n >= 4 && !c->sw.no_mfma_ext && !c->sw.no_hps;
  const bool wide_rec = wide_cand || mfma_small;
  const size_t wide_stride = (size_t)hx::wide_stride(n);
  size_t o_wide = wide_rec ? take((size_t)nt * wide_stride) : 0;
  std::vector<uint64_t> h(off, 0);
  std::vector<uint64_t> mfma_w(wide_rec ? (size_t)nt * n : 0), mfma_negp(wide_rec ? (size_t)nt : 0);   // (mfma_ext.h)
  hxh::BigU P(1);
  for (int k = 0; k < n; k++) {
    h[o_srcq + k] = p[k];
    h[o_srcmu + k] = (uint64_t)((((hxh::u128)1) << 64) / p[k]);
    {
      const double rq = 1.0 / (double)p[k];
      memcpy(&h[o_srcrq + k], &rq, 8);
    }
    for (int l = 0; l < k; l++) {
      uint64_t inv = hxh::invmod(p[l] % p[k], p[k]);
      if (inv == 0)
        return fail(HX_ERR_INVALID, "source primes are not pairwise coprime");
      h[o_ginv + 2 * ((size_t)k * n + l)] = inv;
      h[o_ginv + 2 * ((size_t)k * n + l) + 1] = hxh::shoup(inv, p[k]);
      h[o_ginvm + (size_t)k * n + l] = (uint64_t)(((hxh::u128)inv << 64) % p[k]);
    }
    P.mul_word(p[k]);
  }
  {
    hxh::BigU H = P;  // (P-1)/2 ; P is odd
    H.sub_word(1);
    H.shr1();
    for (int k = 0; k < n; k++)
      h[o_half + k] = H.divmod_word(p[k]);
  }
  uint32_t* tk = reinterpret_cast<uint32_t*>(&h[o_tk]);
  uint32_t* tlazy = reinterpret_cast<uint32_t*>(&h[o_tlazy]);
  uint32_t* tchunk = reinterpret_cast<uint32_t*>(&h[o_tchunk]);
  hxh::u128 sum_src = 0;
  uint64_t max_src = 0, min_src = ~0ull;
  for (int k = 0; k < n; k++) {
    sum_src += p[k];
    max_src = std::max(max_src, p[k]);
    min_src = std::min(min_src, p[k]);
  }
  for (int t = 0; t < nt; t++) {
    uint64_t q = tq(t);
    // lazy 128-bit accumulation is exact when sum_k a_k*W_k < (sum_k q_k)*q_t <= 8*q_t^2
    // (red128_wide's domain; q_t <= 60 bits)
    tlazy[t] = (hxh::bitlen(q) <= 60 && sum_src <= (hxh::u128)8 * q && !c->sw.no_lazy_rns) ? 1u : 0u;
    // seven terms + the carried remainder: r + 7 max_src q < 8 q^2 needs max_src <= q
    tchunk[t] = (hxh::bitlen(q) <= 60 && max_src <= q && !c->sw.no_lazy_rns) ? 1u : 0u;
    h[o_tq + t] = q;
    h[o_tmu64 + t] = (uint64_t)((((hxh::u128)1) << 64) / q);
    int kb = hxh::bitlen(q);
    tk[t] = (uint32_t)kb;
    h[o_tmu + t] = (uint64_t)((((hxh::u128)1) << (2 * kb)) / q);
    h[o_tmu63 + t] = (uint64_t)((((hxh::u128)1) << (63 + kb)) / q);  // < 2^64: q > 2^(kb-1)
    uint64_t run = 1 % q;
    for (int k = 0; k < n; k++) {
      h[o_W + 2 * ((size_t)t * n + k)] = run;
      h[o_W + 2 * ((size_t)t * n + k) + 1] = hxh::shoup(run, q);
      run = hxh::mulmod(run, p[k] % q, q);
    }
    h[o_pmod + t] = run;  // P mod q
    uint64_t pinv = hxh::invmod(run, q);
    h[o_upd + 2 * (size_t)t] = pinv;
    h[o_upd + 2 * (size_t)t + 1] = hxh::shoup(pinv, q);
    if (scaled) {
      if (pinv == 0)
        return fail(HX_ERR_INVALID, "dropped primes are not invertible modulo a kept prime");
      for (int k = 0; k < n; k++) {
        const uint64_t w = hxh::mulmod(h[o_W + 2 * ((size_t)t * n + k)], pinv, q);
        h[o_W + 2 * ((size_t)t * n + k)] = w;
        h[o_W + 2 * ((size_t)t * n + k) + 1] = hxh::shoup(w, q);
      }
      h[o_pmod + t] = 1 % q;
    }
  }
  ExtPlan* pl = new ExtPlan();
  memset(&pl->dev, 0, sizeof pl->dev);
  if (ptxt > 1) {
    uint64_t run = 1 % ptxt;
    for (int k = 0; k < n; k++) {
      h[o_Wp + 2 * (size_t)k] = run;
      h[o_Wp + 2 * (size_t)k + 1] = hxh::shoup(run, ptxt);
      run = hxh::mulmod(run, p[k] % ptxt, ptxt);
    }
    uint64_t pinv = hxh::invmod(run, ptxt);
    if (pinv == 0) {
      delete pl;
      return fail(HX_ERR_INVALID, "dropped primes are not invertible modulo ptxtSpace");
    }
    pl->dev.ptxt = ptxt;
    pl->dev.ptxt_mu64 = (uint64_t)((((hxh::u128)1) << 64) / ptxt);
    int kb = hxh::bitlen(ptxt);
    pl->dev.ptxt_k = (uint32_t)kb;
    pl->dev.ptxt_mu = (uint64_t)((((hxh::u128)1) << (2 * kb)) / ptxt);
    pl->dev.pinv_ptxt = pinv;
    pl->dev.pmod_ptxt = run;
  }
  // Proth-form targets (TgtRec::mont, rns_kernels.h): the record's multipliers, -P mod t (in mu63's slot) and P^-1 mod t
  // (in the slot of 2^64 mod t) times 2^64 -- the fast kernels reduce such a target's limb sum by mont_redc128
  const bool rns_proth = !c->sw.no_proth && !c->sw.no_proth_rns && n <= 16;
  auto tgt_mont = [&](int t) { return rns_proth && hx::is_proth32(tq(t)); };
  auto rec_to_mont = [&](uint64_t* rec) {
    const uint64_t q = rec[0];
    auto m64 = [&](uint64_t x) { return (uint64_t)(((hxh::u128)(x % q) << 64) % q); };
    rec[2] = m64(q - rec[1] % q);
    rec[4] |= (uint64_t)1 << 10;
    for (int k = 0; k < n; k++)
      rec[8 + k] = m64(rec[8 + k]);
    rec[8 + 2 * n] = m64(rec[5]);
  };
  for (int t = 0; t < nt; t++) {
    uint64_t* rec = &h[o_pack + (size_t)t * pack_stride];
    rec[0] = h[o_tq + t];
    rec[1] = h[o_pmod + t];
    rec[2] = h[o_tmu63 + t];
    rec[3] = h[o_tmu64 + t];
    rec[4] = (uint64_t)tk[t] | ((uint64_t)tlazy[t] << 8) | ((uint64_t)tchunk[t] << 9);
    rec[5] = h[o_upd + 2 * (size_t)t];
    rec[6] = h[o_upd + 2 * (size_t)t + 1];
    rec[7] = h[o_tmu + t];
    for (int k = 0; k < n; k++) {
      rec[8 + k] = h[o_W + 2 * ((size_t)t * n + k)];
      rec[8 + n + k] = h[o_W + 2 * ((size_t)t * n + k) + 1];
    }
    const uint64_t r64 = (uint64_t)((((hxh::u128)1) << 64) % rec[0]);   // red128_any's constant
    rec[8 + 2 * n] = r64;
    rec[8 + 2 * n + 1] = hxh::shoup(r64, rec[0]);
    if (tgt_mont(t))
      rec_to_mont(rec);
  }
  // HPS front end: y_k = a_k (P/p_k)^-1 mod p_k, multipliers (P/p_k) mod t (scaled plans: / P, i.e.
  // p_k^-1 mod t), the same header; "lazy" needs room for up to n + 1 extra multiples of t in the sum
  bool hps_ok = n >= 2 && (n <= 16 || wide_cand) && (ptxt <= 1 || ptxt < ((uint64_t)1 << (wide_cand ? 56 : 58))) &&
                (!c->sw.no_hps || wide_cand);
  if (hps_ok) {
    auto prod_except = [&](int k, uint64_t m) {   // (P / p_k) mod m
      uint64_t r = 1 % m;
      for (int l = 0; l < n; l++)
        if (l != k)
          r = hxh::mulmod(r, p[l] % m, m);
      return r;
    };
    for (int k = 0; k < n && hps_ok; k++) {
      const uint64_t inv = hxh::invmod(prod_except(k, p[k]), p[k]);
      hps_ok = inv != 0;
      h[o_hinv + 2 * (size_t)k] = inv;
      h[o_hinv + 2 * (size_t)k + 1] = hxh::shoup(inv, p[k]);
      if (ptxt > 1) {
        const uint64_t w = prod_except(k, ptxt);
        h[o_Wp2 + 2 * (size_t)k] = w;
        h[o_Wp2 + 2 * (size_t)k + 1] = hxh::shoup(w, ptxt);
      }
    }
    for (int t = 0; t < nt && hps_ok && wide_rec; t++) {
      // WideRec: q, P mod t (scaled: 1), 2^64 mod t with its Shoup companion, floor(2^64/t), P^-1 mod t with its
      // companion, the companion of P mod t; then the multipliers (P/p_k) mod t (scaled: / P) as 30-bit limbs
      const uint64_t q = tq(t);
      uint64_t* rec = &h[o_wide + (size_t)t * wide_stride];
      const uint64_t r64 = (uint64_t)((((hxh::u128)1) << 64) % q);
      rec[0] = q;
      rec[1] = h[o_pmod + t];
      rec[2] = r64;
      rec[3] = hxh::shoup(r64, q);
      rec[4] = h[o_tmu64 + t];
      rec[5] = h[o_upd + 2 * (size_t)t];
      rec[6] = h[o_upd + 2 * (size_t)t + 1];
      rec[7] = hxh::shoup(h[o_pmod + t], q);
      const uint64_t pinv_t = h[o_upd + 2 * (size_t)t];   // P^-1 mod t
      for (int k = 0; k < n; k++) {
        uint64_t w = prod_except(k, q);
        if (scaled)
          w = hxh::mulmod(w, pinv_t, q);
        rec[8 + k] = (w & 0x3fffffffull) | ((w >> 30) << 32);
        mfma_w[(size_t)t * n + k] = w;
      }
      mfma_negp[t] = (q - rec[1] % q) % q;
    }
    for (int t = 0; t < nt && hps_ok && !wide_cand; t++) {
      const uint64_t q = tq(t);
      const uint64_t* rec = &h[o_pack + (size_t)t * pack_stride];
      uint64_t* rec2 = &h[o_pack2 + (size_t)t * pack_stride];
      for (int j = 0; j < 8; j++)
        rec2[j] = rec[j];   // (a Proth-form target: rec[2] is -P 2^64 already)
      rec2[8 + 2 * n] = rec[8 + 2 * n];
      rec2[8 + 2 * n + 1] = rec[8 + 2 * n + 1];
      const uint32_t lazy2 = (hxh::bitlen(q) <= 60 && sum_src + 32 <= (hxh::u128)8 * q && !c->sw.no_lazy_rns) ? 1u : 0u;
      rec2[4] = (uint64_t)tk[t] | ((uint64_t)lazy2 << 8) | ((uint64_t)tchunk[t] << 9) | (rec[4] & ((uint64_t)1 << 10));
      const uint64_t pinv_t = h[o_upd + 2 * (size_t)t];   // P^-1 mod t
      for (int k = 0; k < n; k++) {
        uint64_t w = prod_except(k, q);
        if (scaled)
          w = hxh::mulmod(w, pinv_t, q);
        rec2[8 + n + k] = hxh::shoup(w, q);
        if (tgt_mont(t))
          w = (uint64_t)(((hxh::u128)w << 64) % q);
        rec2[8 + k] = w;
      }
    }
  }
  uint64_t* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, off * 8));
  HIPCHK(hipMemcpy(d, h.data(), off * 8, hipMemcpyHostToDevice));
  pl->blob = d;
  pl->dev.tgt_pack = hx::as_ro(d + o_pack);
  pl->dev.ginv_m = hx::as_ro(d + o_ginvm);
  {
    bool sm = rns_proth;
    for (int k = 0; k < n && sm; k++)
      sm = hx::is_proth32(p[k]);
    pl->dev.src_mont = sm ? 1u : 0u;
  }
  pl->dev.tgt_pack_hps = hx::as_ro(d + o_pack2);
  pl->dev.hps_inv = hx::as_ro(reinterpret_cast<const TW*>(d + o_hinv));
  pl->dev.Wp_hps = hx::as_ro(reinterpret_cast<const TW*>(d + o_Wp2));
  pl->dev.hps_eps = c->sw.hps_eps;
  pl->dev.n = n;
  pl->dev.nt = nt;
  pl->dev.src_q = hx::as_ro(d + o_srcq);
  pl->dev.src_mu64 = hx::as_ro(d + o_srcmu);
  pl->dev.ginv = hx::as_ro(reinterpret_cast<const TW*>(d + o_ginv));
  pl->dev.half = hx::as_ro(d + o_half);
  pl->dev.tgt_q = hx::as_ro(d + o_tq);
  pl->dev.tgt_mu64 = hx::as_ro(d + o_tmu64);
  pl->dev.tgt_mu = hx::as_ro(d + o_tmu);
  pl->dev.tgt_k = hx::as_ro(reinterpret_cast<const uint32_t*>(d + o_tk));
  pl->dev.pmod = hx::as_ro(d + o_pmod);
  pl->dev.W = hx::as_ro(reinterpret_cast<const TW*>(d + o_W));
  pl->dev.upd = hx::as_ro(reinterpret_cast<const TW*>(d + o_upd));
  pl->dev.Wp = hx::as_ro(reinterpret_cast<const TW*>(d + o_Wp));
  pl->dev.tgt_lazy = hx::as_ro(reinterpret_cast<const uint32_t*>(d + o_tlazy));
  // a_l < q_l <= max < 2*min <= 2*p_k, or the sources ascend (a_l < p_l <= p_k for l < k)
  pl->dev.garner_cs = ((max_src / 2 < min_src || std::is_sorted(p.begin(), p.end())) && !c->sw.no_lazy_rns) ? 1u : 0u;
  pl->dev.src_rq = hx::as_ro(reinterpret_cast<const double*>(d + o_srcrq));
  pl->dev.tgt_mu63 = hx::as_ro(d + o_tmu63);
  {
    bool ok = pl->dev.garner_cs && n <= 8 && (min_src >> 32) != 0 && !c->sw.no_fast_break;
    for (int t = 0; t < nt && ok; t++)
      ok = (tq(t) >> 32) != 0 && hxh::bitlen(tq(t)) <= 60;
    pl->dev.fast_ok = ok ? 1u : 0u;
    bool ok16 = pl->dev.garner_cs && n <= 16 && (min_src >> 32) != 0 && !c->sw.no_fast_extend;
    for (int t = 0; t < nt && ok16; t++)
      ok16 = (tq(t) >> 32) != 0 && hxh::bitlen(tq(t)) <= 60;
    pl->dev.fast16_ok = ok16 ? 1u : 0u;
    {
      // lazy output is read at the Proth rows' tight bounds: legal only when the kernel that wrote a Proth-form
      // target row took its Montgomery branch (tgt_mont) -- derived here from the plan's own flags, not from the
      // environment switches the call sites used to consult
      bool lz = hxh::bitlen(max_src) <= 60;
      for (int t = 0; t < nt && lz; t++)
        if (!c->sw.no_proth && hx::is_proth32(tq(t)) && !tgt_mont(t))
          lz = false;
      pl->dev.lazy_tight_ok = lz ? 1u : 0u;
    }
    pl->dev.hps_ok = (hps_ok && ok16) ? 1u : 0u;   // (the front end lives in the fast kernels only)
    bool okw = wide_cand && hps_ok && (min_src >> 32) != 0 && hxh::bitlen(max_src) <= 60;
    for (int t = 0; t < nt && okw; t++)
      okw = (tq(t) >> 32) != 0 && hxh::bitlen(tq(t)) <= 60;
    pl->dev.wide_ok = okw ? 1u : 0u;
    pl->dev.wide_pack = hx::as_ro(d + o_wide);
    // the same extension on the matrix cores: multipliers as balanced 8-bit limbs in MFMA operand order (mfma_ext.h)
    pl->dev.mfma_a = nullptr;
    pl->dev.mfma_steps = 0;
    const bool oks = mfma_small && pl->dev.hps_ok;   // (hps_ok: the fast kernels' preconditions hold and the HPS tables exist)
    if ((okw || oks) && !c->sw.no_mfma_ext) {
      std::vector<uint8_t> tab;
      std::vector<uint64_t> tqs((size_t)nt);
      for (int t = 0; t < nt; t++)
        tqs[(size_t)t] = tq(t);
      hx::mfx::build_tables(n, nt, tqs.data(), mfma_w.data(), mfma_negp.data(), &h[o_upd], tab);
      uint8_t* dm = nullptr;
      HIPCHK(hipMalloc((void**)&dm, tab.size()));
      HIPCHK(hipMemcpy(dm, tab.data(), tab.size(), hipMemcpyHostToDevice));
      pl->blob_mfma = dm;
      pl->dev.mfma_a = dm;
      pl->dev.mfma_steps = (uint32_t)hx::mfx::steps_for(n);
    }
  }
  pl->dev.tgt_chunk7 = hx::as_ro(reinterpret_cast<const uint32_t*>(d + o_tchunk));
  {
    bool unit = scaled && ptxt > 1;   // (P mod t is 1 in a scaled plan)
    for (int t = 0; t < nt && unit; t++)
      unit = ptxt <= tq(t);
    pl->dev.corr_unit = unit ? 1u : 0u;
  }
  c->plans[key] = pl;
  *out = pl;
  return HX_OK;
}

// The redo list of the HPS-form RNS kernels (rns_kernels.h: ExtRep): room for every coefficient of the launch, count
// zeroed on the stream in front of the launch that fills it.
static int redo_prepare(hx_ctx* c, size_t row_words, uint32_t** out)
{
  const size_t need = row_words + 1;
  if (c->redo_cap < need) {
    uint32_t* nd = nullptr;
    HIPCHK(hipMalloc((void**)&nd, need * sizeof(uint32_t)));
    retire_or_free(c, c->d_redo);
    c->d_redo = nd;
    c->redo_cap = need;
  }
  HIPCHK(hipMemsetAsync(c->d_redo, 0, sizeof(uint32_t), c->stream));
  *out = c->d_redo;
  return HX_OK;
}
// The HPS form pays from nine source primes on: measured same-box (gpurun_out r3c29), the digit kernel with 5-8 primes
// per digit is 3-5 % SLOWER with it (366 vs 350 us at BGV L=16, 828 vs 808 us at CKKS L=24: eight u64 -> double
// conversions cost what the 10-28 dependent Garner products cost), the basis extension of 11 dropped primes is faster
// (Garner there is 55 products and spills 62 dwords; CKKS level 2 +3 %).  HX_HPS_MIN_N overrides (tests force 2).
static int hps_min_n(const hx_ctx* c)
{
  return c->sw.hps_min_n;
}
static const dim3 REDO_GRID(64);   // the Garner pass over the listed coefficients (grid-stride; the list is almost always empty)

static int launch_extend(hx_ctx* c, const ExtPlan* pl, const ExtArgs& args_in, size_t row_words)
{
  dim3 grid((unsigned)((row_words + 255) / 256)), block(256);
  int n = pl->dev.n;
  ExtArgs args = args_in;
  args.redo = nullptr;
  if (pl->dev.fast16_ok) {
    // HPS form + Garner over its redo list when the plan has the tables and no row is updated in place (an in-place
    // update cannot be redone); otherwise Garner over everything
    // (the plan carries the matrix-core tables from mfma_min_n sources on: below hps_min_n too)
    const bool mfma = pl->dev.hps_ok && pl->dev.mfma_steps != 0 && args.upd == nullptr && row_words < ((size_t)1 << 32);
    const bool hps = mfma || (pl->dev.hps_ok && n >= hps_min_n(c) && args.upd == nullptr && row_words < ((size_t)1 << 32));
    if (hps)
      CHK(redo_prepare(c, row_words, &args.redo));
    if (mfma)   // the HPS form with its target sums on the matrix cores; the Garner form below over its redo list
      HIPCHK(hx::launch_rns_extend_mfma(pl->dev, args, row_words, c->stream));
#define HX_EXT_FAST(NN)                                                                                            \
  case NN:                                                                                                         \
    if (hps) {                                                                                                     \
      if (!mfma)                                                                                                   \
        HX_LAUNCH((hx::rns_extend_fast_kernel<NN, true>), grid, block, 0, c->stream, pl->dev, args, row_words);    \
      HX_LAUNCH((hx::rns_extend_fast_kernel<NN, false>), REDO_GRID, block, 0, c->stream, pl->dev, args, row_words); \
    } else {                                                                                                       \
      HX_LAUNCH((hx::rns_extend_fast_kernel<NN, false>), grid, block, 0, c->stream, pl->dev, args, row_words);     \
    }                                                                                                              \
    break;
    switch (n) {
      HX_EXT_FAST(1) HX_EXT_FAST(2) HX_EXT_FAST(3) HX_EXT_FAST(4) HX_EXT_FAST(5) HX_EXT_FAST(6) HX_EXT_FAST(7) HX_EXT_FAST(8)
      HX_EXT_FAST(9) HX_EXT_FAST(10) HX_EXT_FAST(11) HX_EXT_FAST(12) HX_EXT_FAST(13) HX_EXT_FAST(14) HX_EXT_FAST(15)
      HX_EXT_FAST(16)
    }
#undef HX_EXT_FAST
    HIPCHK(hipGetLastError());
    return HX_OK;
  }
  // (a plan whose multiplier table does not fit the LDS goes to the generic kernel below, which handles any plan)
  const size_t wide_lds = (size_t)pl->dev.nt * (size_t)hx::wide_stride(n) * 8;   // the plan's multipliers, once per workgroup
  if (pl->dev.wide_ok && row_words < ((size_t)1 << 32) && wide_lds <= 160 * 1024) {
    // 17..40 source primes (the reference's own benchmark chain): HPS form, then Garner over the coefficients it
    // could not vouch for (they were left untouched, so in-place updates are redone correctly too)
    CHK(redo_prepare(c, row_words, &args.redo));
    if (pl->dev.mfma_steps) {
      // the target sums as an int8 matrix product (rns_mfma_kernels.hip), the Garner pass over its redo list behind it
      HIPCHK(hx::launch_rns_extend_mfma(pl->dev, args, row_words, c->stream));
      HX_LAUNCH((hx::rns_extend_kernel<40>), REDO_GRID, block, 0, c->stream, pl->dev, args, row_words);
      HIPCHK(hipGetLastError());
      return HX_OK;
    }
    const dim3 wgrid((unsigned)((row_words + hx::WIDE_THREADS - 1) / hx::WIDE_THREADS)), wblock(hx::WIDE_THREADS);
    const size_t lds = wide_lds;
    // (function attributes are per device: one flag per device a context of this process has used)
    static std::mutex wide_mu;
    static std::unordered_set<int> wide_attr;
    {
      std::lock_guard<std::mutex> lk(wide_mu);
      if (!wide_attr.count(c->device)) {
        for (const void* f : {(const void*)hx::rns_extend_wide_kernel<20>, (const void*)hx::rns_extend_wide_kernel<24>,
                              (const void*)hx::rns_extend_wide_kernel<28>, (const void*)hx::rns_extend_wide_kernel<32>,
                              (const void*)hx::rns_extend_wide_kernel<36>, (const void*)hx::rns_extend_wide_kernel<40>})
          HIPCHK(hxp::dyn_lds(f, 160 * 1024));
        wide_attr.insert(c->device);
      }
    }
    if (n <= 20)
      HX_LAUNCH((hx::rns_extend_wide_kernel<20>), wgrid, wblock, lds, c->stream, pl->dev, args, row_words);
    else if (n <= 24)
      HX_LAUNCH((hx::rns_extend_wide_kernel<24>), wgrid, wblock, lds, c->stream, pl->dev, args, row_words);
    else if (n <= 28)
      HX_LAUNCH((hx::rns_extend_wide_kernel<28>), wgrid, wblock, lds, c->stream, pl->dev, args, row_words);
    else if (n <= 32)
      HX_LAUNCH((hx::rns_extend_wide_kernel<32>), wgrid, wblock, lds, c->stream, pl->dev, args, row_words);
    else if (n <= 36)
      HX_LAUNCH((hx::rns_extend_wide_kernel<36>), wgrid, wblock, lds, c->stream, pl->dev, args, row_words);
    else
      HX_LAUNCH((hx::rns_extend_wide_kernel<40>), wgrid, wblock, lds, c->stream, pl->dev, args, row_words);
    HX_LAUNCH((hx::rns_extend_kernel<40>), REDO_GRID, block, 0, c->stream, pl->dev, args, row_words);
    HIPCHK(hipGetLastError());
    return HX_OK;
  }
  if (n <= 8)
    HX_LAUNCH((hx::rns_extend_kernel<8>), grid, block, 0, c->stream, pl->dev, args,
                       row_words);
  else if (n <= 16)
    HX_LAUNCH((hx::rns_extend_kernel<16>), grid, block, 0, c->stream, pl->dev, args,
                       row_words);
  else if (n <= 40)
    HX_LAUNCH((hx::rns_extend_kernel<40>), grid, block, 0, c->stream, pl->dev, args,
                       row_words);
  else if (n <= 64)
    HX_LAUNCH((hx::rns_extend_kernel<64>), grid, block, 0, c->stream, pl->dev, args,
                       row_words);
  else  // whole chains of the reference's own benchmark parameter (bits=6400: 143 primes, e.g. the
        // toPoly of a decryption): the digits live in private memory -- slow, and rare
    HX_LAUNCH((hx::rns_extend_kernel<MAX_EXT_SRC>), grid, block, 0, c->stream, pl->dev, args,
                       row_words);
  HIPCHK(hipGetLastError());
  return HX_OK;
}

static void clear_args(ExtArgs& a)
{
  memset(&a, 0xff, sizeof a);
  a.src = nullptr;
  a.dst = nullptr;
  a.upd = nullptr;
  a.nu = 0;
  a.frac = nullptr;
  a.redo = nullptr;
  a.lazy_out = 0;
}

// ------------------------------------------------------------------
// canonical-embedding norms on the device (src/norms.cpp:129-262)
// ------------------------------------------------------------------
// Arms the "fraction" side output of the exact-RNS kernels: `doubles` values of room.
// size of the complex Bluestein transform of the general-m norms: the integer transform's 2^bk
static int bn_logp(const hx_ctx* c) { return next_pow2_exp(2 * c->m - 1); }

static int frac_begin(hx_ctx* c, size_t doubles)
{
  if (!c->pow2 && bn_logp(c) > 18)
    return fail(HX_ERR_UNSUPPORTED,
                "device embedding norms need m a power of two or m <= 131072 (otherwise the host keeps "
                "the reference's noiseBoundForUniform bound)");
  if (c->frac_cap < doubles) {
    CHK(norm_join(c, false, true));
    retire_or_free(c, c->d_frac);
    c->d_frac = nullptr;
    c->frac_cap = 0;
    HIPCHK(hipMalloc((void**)&c->d_frac, doubles * sizeof(double)));
    c->frac_cap = doubles;
  }
  CHK(norm_join(c, false, true));  // d_frac is about to be rewritten
  c->frac_pos = 0;
  c->want_frac = true;
  c->xs_rows = 0;
  return HX_OK;
}
static double* frac_take(hx_ctx* c, size_t doubles)
{
  if (!c->want_frac || c->frac_pos + doubles > c->frac_cap)
    return nullptr;
  double* p = c->d_frac + c->frac_pos;
  c->frac_pos += doubles;
  return p;
}

// A fused mod-down block whose (x, S) the norm kernel was going to read in place: write its
// fdelta out now (something else is about to reuse the scratch slots, or the batch is mixed).
static int flush_xs(hx_ctx* c)
{
  if (c->xs_rows <= 0)
    return HX_OK;
  const size_t n = (size_t)c->xs_rows * c->phim;
  HX_LAUNCH(hx::frac_from_xs_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)),
                     dim3(256), 0, c->stream, c->scratch[0],
                     reinterpret_cast<const int64_t*>(c->scratch[1]), c->xs_inv_qd,
                     c->d_frac + (size_t)c->xs_first * c->phim, n);
  HIPCHK(hipGetLastError());
  c->xs_rows = 0;
  return HX_OK;
}

// out_host[r] = max_j |f_r(W^(2j+1))| for `rows` real polynomials of N coefficients at d_f.
// Synchronises the stream (the caller needs the numbers on the host).
// general m: tables of the complex-double Bluestein transform (norm_kernels.h), made on first use
static int bnorm_setup(hx_ctx* c)
{
  if (c->d_bn_chat)
    return HX_OK;
  const int bk = bn_logp(c);
  if (bk > 18)
    return fail(HX_ERR_UNSUPPORTED, "device embedding norms: m too large (2m-1 > 2^18)");
  const uint64_t m = c->m;
  const size_t P = (size_t)1 << bk;
  const long double pi = 3.141592653589793238462643383279502884L;
  std::vector<double> v(2 * m), cv(2 * P, 0.0), w(P);
  for (uint64_t k = 0; k < m; k++) {
    const uint64_t k2 = (uint64_t)(((hxh::u128)k * k) % (2 * m));  // exact angle reduction
    const long double ang = pi * (long double)k2 / (long double)m;
    const double cr = (double)cosl(ang), si = (double)sinl(ang);
    v[2 * k] = cr;
    v[2 * k + 1] = si;
    cv[2 * k] = cr;  // c_k = conj(v_k), at k and at P-k
    cv[2 * k + 1] = -si;
    if (k) {
      cv[2 * (P - k)] = cr;
      cv[2 * (P - k) + 1] = -si;
    }
  }
  for (size_t k = 0; k < P / 2; k++) {
    const long double ang = 2 * pi * (long double)k / (long double)P;
    w[2 * k] = (double)cosl(ang);
    w[2 * k + 1] = (double)sinl(ang);
  }
  double2* d_cv = nullptr;
  HIPCHK(hipMalloc((void**)&c->d_bn_v, sizeof(double) * 2 * m));
  HIPCHK(hipMalloc((void**)&c->d_bn_w, sizeof(double) * P));
  HIPCHK(hipMalloc((void**)&c->d_bn_chat, sizeof(double) * 2 * P));
  HIPCHK(hipMalloc((void**)&d_cv, sizeof(double) * 2 * P));
  HIPCHK(hipMemcpy(c->d_bn_v, v.data(), sizeof(double) * 2 * m, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c->d_bn_w, w.data(), sizeof(double) * P, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d_cv, cv.data(), sizeof(double) * 2 * P, hipMemcpyHostToDevice));
  bool attr = false;   // (hxp::dyn_lds is idempotent per device)
  if (!attr) {
    HIPCHK(hxp::dyn_lds((const void*)hx::bnorm_fwd_kernel, 16 << hx::NORM_MAX_LOGH));
    HIPCHK(hxp::dyn_lds((const void*)hx::bnorm_inv_kernel, 16 << hx::NORM_MAX_LOGH));
    attr = true;
  }
  const int logp = bk, logh = std::min(logp, hx::NORM_MAX_LOGH);
  const unsigned H = 1u << logh, S = (unsigned)(P >> logh);
  const unsigned threads = std::min<unsigned>(hx::NORM_THREADS, std::max<unsigned>(64u, H / 4));
  HX_LAUNCH(hx::bnorm_fwd_kernel, dim3(S), dim3(threads), std::max<size_t>(16 * (size_t)H, 256), c->stream,
                     (const double*)nullptr, (const double2*)nullptr, (const double2*)d_cv, c->d_bn_w,
                     (const double2*)nullptr, c->d_bn_chat, logp, logh, c->phim, 1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipFree(d_cv));
  return HX_OK;
}

// squared norms of `rows` polynomials into c->d_norm2 (general m)
static int embed_norms_general(hx_ctx* c, const double* d_f, int rows, hipStream_t ns)
{
  CHK(bnorm_setup(c));
  const int logp = bn_logp(c), logh = std::min(logp, hx::NORM_MAX_LOGH);
  const size_t P = (size_t)1 << logp;
  const unsigned H = 1u << logh, S = (unsigned)(P >> logh);
  const size_t cap = std::max<size_t>(1, ((size_t)1 << 28) / (16 * P));  // <= 256 MiB of work buffer
  if (c->bn_rows_cap < std::min<size_t>(cap, (size_t)rows)) {
    retire_or_free(c, c->d_bn_Z);
    c->d_bn_Z = nullptr;
    c->bn_rows_cap = std::min<size_t>(cap, (size_t)rows);
    HIPCHK(hipMalloc((void**)&c->d_bn_Z, c->bn_rows_cap * P * 16));
  }
  const unsigned threads = std::min<unsigned>(hx::NORM_THREADS, std::max<unsigned>(64u, H / 4));
  const size_t lds = std::max<size_t>(16 * (size_t)H, 256);
  for (size_t r0 = 0; r0 < (size_t)rows; r0 += c->bn_rows_cap) {
    const unsigned nr = (unsigned)std::min<size_t>(c->bn_rows_cap, (size_t)rows - r0);
    HX_LAUNCH(hx::bnorm_fwd_kernel, dim3(nr * S), dim3(threads), lds, ns,
                       d_f + r0 * c->phim, (const double2*)c->d_bn_v, (const double2*)nullptr, c->d_bn_w,
                       (const double2*)c->d_bn_chat, c->d_bn_Z, logp, logh, c->phim, 0);
    HIPCHK(hipGetLastError());
    HX_LAUNCH(hx::bnorm_inv_kernel, dim3(nr * S), dim3(threads), lds, ns, c->d_bn_Z, c->d_bn_w,
                       logp, logh);
    HIPCHK(hipGetLastError());
    const unsigned gx = std::min<unsigned>(64u, (c->phim + 255u) / 256u);
    HX_LAUNCH(hx::bnorm_max_kernel, dim3(gx, nr), dim3(256), 0, ns, (const double2*)c->d_bn_Z,
                       c->d_bn_w, c->d_zms, c->phim, logp, logh, c->d_norm2 + r0);
    HIPCHK(hipGetLastError());
  }
  return HX_OK;
}

static int embed_norms(hx_ctx* c, const double* d_f, int rows, double* out_host)
{
  const uint32_t N = c->phim;
  const int logn = c->logn;
  if (c->pow2 && !c->d_wtab) {
    std::vector<double> h(2 * (size_t)N);
    const long double two_pi = 6.283185307179586476925286766559005768394L;
    for (uint32_t k = 0; k < N; k++) {
      long double ang = two_pi * (long double)k / (long double)c->m;
      h[2 * (size_t)k] = (double)cosl(ang);
      h[2 * (size_t)k + 1] = (double)sinl(ang);
    }
    HIPCHK(hipMalloc((void**)&c->d_wtab, sizeof(double) * 2 * (size_t)N));
    HIPCHK(hipMemcpy(c->d_wtab, h.data(), sizeof(double) * 2 * (size_t)N, hipMemcpyHostToDevice));
  }
  if (c->norm_cap < (size_t)rows) {
    retire_or_free(c, c->d_norm2);
    c->d_norm2 = nullptr;
    c->norm_cap = 0;
    HIPCHK(hipMalloc((void**)&c->d_norm2, sizeof(unsigned long long) * (size_t)rows));
    c->norm_cap = (size_t)rows;
  }
  // which source the kernel will read: the mod-down's (x, S) in place, or doubles at d_f (then a pending
  // (x, S) block is written out as doubles first -- on the main stream, before the hand-over below)
  const bool quarter_form = c->pow2 && logn >= 2 && logn - 1 <= hx::NORM_MAX_LOGH;
  const bool use_xs = quarter_form && c->xs_rows == rows && d_f == c->d_frac;
  if (!use_xs)
    CHK(flush_xs(c));
  hipStream_t ns = c->stream;
  // (measured, profiles/r03_norm_side_stream_ab.txt: no gain -- the one-workgroup-per-CU norm kernel does not get
  // onto the CUs next to a kernel that fills them -- so the side stream is opt-in: HX_NORM_ASYNC=1)
  const bool side_stream = c->sw.norm_async;
  if (side_stream && !c->capturing) {
    if (!c->norm_stream) {
      HIPCHK(hipStreamCreateWithFlags(&c->norm_stream, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&c->norm_in_ev, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&c->norm_out_ev, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(c->norm_in_ev, c->stream));             // behind the producers of the input
    HIPCHK(hipStreamWaitEvent(c->norm_stream, c->norm_in_ev, 0));
    ns = c->norm_stream;
  }
  // read-back through a pinned slot and an event
  hx_ctx::NormPending np{};
  bool have = false;
  for (size_t i = 0; i < c->norm_free.size(); i++)
    if (c->norm_free[i].cap >= (size_t)rows) {
      np = c->norm_free[i];
      c->norm_free.erase(c->norm_free.begin() + (long)i);
      have = true;
      break;
    }
  if (!have && rows <= 1024) {
    constexpr size_t PER = 64, CAP = 1024;
    unsigned long long* slab = nullptr;
    HIPCHK(hipHostMalloc((void**)&slab, PER * CAP * sizeof(unsigned long long), hipHostMallocDefault));
    c->norm_slabs.push_back(slab);
    for (size_t i = 0; i < PER; i++) {
      hx_ctx::NormPending f{};
      f.pinned = slab + i * CAP;
      f.cap = CAP;
      f.slab = true;
      HIPCHK(hipEventCreateWithFlags(&f.ev, hipEventDisableTiming));
      c->norm_free.push_back(f);
    }
    np = c->norm_free.back();
    c->norm_free.pop_back();
    have = true;
  }
  if (!have) {
    np.cap = (size_t)rows;
    np.slab = false;
    HIPCHK(hipHostMalloc((void**)&np.pinned, np.cap * sizeof(unsigned long long), hipHostMallocDefault));
    HIPCHK(hipEventCreateWithFlags(&np.ev, hipEventDisableTiming));
  }
  // (the slot goes back to the free list if anything below fails before it is queued)
  struct SlotGuard {
    hx_ctx* c;
    hx_ctx::NormPending* np;
    bool armed = true;
    ~SlotGuard()
    {
      if (armed)
        c->norm_free.push_back(*np);
    }
  } slot_guard{c, &np};
  // One workgroup per polynomial and a plain store of its maximum (the radix-16 kernels): the kernel writes the pinned,
  // device-visible slot itself -- no zero-fill of d_norm2 in front of it (an atomicMax target) and no copy kernel behind
  // it, two ~4 us launches with their gaps per norm call, three calls per multiply.
  const bool by_copy = c->sw.norm_memcpy;
  const bool r16_path = c->pow2 && !c->sw.norm_old && logn == 14;
  const bool x2_path = c->pow2 && logn == 15 && !c->sw.norm_old && !c->sw.norm_plain;
  const bool direct = !by_copy && (r16_path || x2_path);
  unsigned long long* const out2 = direct ? np.pinned : c->d_norm2;
  if (!direct)
    HIPCHK(hipMemsetAsync(c->d_norm2, 0, sizeof(unsigned long long) * (size_t)rows, ns));
  bool attr = false;   // (hxp::dyn_lds is idempotent per device)
  if (!c->pow2) {
    CHK(flush_xs(c));
    CHK(embed_norms_general(c, d_f, rows, ns));
  } else if (!attr) {
    HIPCHK(hxp::dyn_lds((const void*)hx::embed_norm_kernel, 16 << hx::NORM_MAX_LOGH));
#define HX_NORM_ATTR(...)                                                                               \
  HIPCHK(hxp::dyn_lds((const void*)hx::embed_norm_quarter_kernel<__VA_ARGS__>,                   \
                             hipFuncAttributeMaxDynamicSharedMemorySize, 16 << hx::NORM_MAX_LOGH))
    HX_NORM_ATTR(hx::NormSrcF64, 0);
    HX_NORM_ATTR(hx::NormSrcF64, 13);
    HX_NORM_ATTR(hx::NormSrcF64, 14);
    HX_NORM_ATTR(hx::NormSrcXS, 0);
    HX_NORM_ATTR(hx::NormSrcXS, 13);
    HX_NORM_ATTR(hx::NormSrcXS, 14);
#undef HX_NORM_ATTR
    HIPCHK(hxp::dyn_lds((const void*)hx::embed_norm_quarter_split_kernel, 16 << hx::NORM_MAX_LOGH));
    attr = true;
  }
  if (!c->pow2) {
    // (done above)
  } else if (logn >= 2 && logn - 1 <= hx::NORM_MAX_LOGH) {
    // real-input form: one N/2-point transform per polynomial, one workgroup each
    const unsigned M = N >> 1;
    const unsigned threads = std::min<unsigned>(hx::NORM_THREADS, std::max<unsigned>(64u, M / 4));
    const size_t lds = std::max<size_t>(16 * (size_t)M, 256);
    // (N = 2^13 / 2^14 with the full thread count: the instantiations with constant loop bounds)
    const bool full = threads == (unsigned)hx::NORM_THREADS;
#define HX_NORM_LAUNCH(SRCT, srcv)                                                                              \
  do {                                                                                                          \
    if (full && logn == 14)                                                                                     \
      HX_LAUNCH((hx::embed_norm_quarter_kernel<SRCT, 14>), dim3((unsigned)rows), dim3(threads), lds,   \
                         ns, srcv, c->d_wtab, logn, c->d_norm2);                                         \
    else if (full && logn == 13)                                                                                \
      HX_LAUNCH((hx::embed_norm_quarter_kernel<SRCT, 13>), dim3((unsigned)rows), dim3(threads), lds,   \
                         ns, srcv, c->d_wtab, logn, c->d_norm2);                                         \
    else                                                                                                        \
      HX_LAUNCH((hx::embed_norm_quarter_kernel<SRCT, 0>), dim3((unsigned)rows), dim3(threads), lds,    \
                         ns, srcv, c->d_wtab, logn, c->d_norm2);                                         \
  } while (0)
    // N = 2^14: the register-tiled kernel (norm_r16.h); HX_NORM_OLD keeps the LDS-pass kernel (A/B)
    const bool r16 = !c->sw.norm_old;
    constexpr size_t r16_lds = (size_t)hx::R16_LDS_DOUBLES * sizeof(double);   // one array: two workgroups per CU
    if (r16 && logn == 14) {
      bool attr16 = false;   // (hxp::dyn_lds is idempotent per device)
      if (!attr16) {
        HIPCHK(hxp::dyn_lds((const void*)hx::embed_norm_r16_kernel<hx::NormSrcXS>, (int)r16_lds));
        HIPCHK(hxp::dyn_lds((const void*)hx::embed_norm_r16_kernel<hx::NormSrcF64>, (int)r16_lds));
        attr16 = true;
      }
    }
    if (c->xs_rows == rows && d_f == c->d_frac) {
      hx::NormSrcXS src{c->scratch[0], reinterpret_cast<const int64_t*>(c->scratch[1]), c->xs_inv_qd};
      if (r16 && logn == 14)
        HX_LAUNCH((hx::embed_norm_r16_kernel<hx::NormSrcXS>), dim3((unsigned)rows), dim3(hx::R16_THREADS), r16_lds, ns, src,
                  c->d_wtab, out2, direct);
      else
        HX_NORM_LAUNCH(hx::NormSrcXS, src);
    } else {
      CHK(flush_xs(c));
      hx::NormSrcF64 src{d_f};
      if (r16 && logn == 14)
        HX_LAUNCH((hx::embed_norm_r16_kernel<hx::NormSrcF64>), dim3((unsigned)rows), dim3(hx::R16_THREADS), r16_lds, ns, src,
                  c->d_wtab, out2, direct);
      else
        HX_NORM_LAUNCH(hx::NormSrcF64, src);
    }
#undef HX_NORM_LAUNCH
    c->xs_rows = 0;
  } else if (logn - 1 > hx::NORM_MAX_LOGH && !c->sw.norm_plain) {
    // real-input form beyond one workgroup's LDS: S = N/2/8192 sub-transforms, one workgroup per pair
    CHK(flush_xs(c));
    const int logh = hx::NORM_MAX_LOGH;
    const unsigned H = 1u << logh, S = (N >> 1) >> logh;
    const size_t park_words = (size_t)rows * (S / 2) * H;   // complex doubles
    const bool x2 = logn == 15 && !c->sw.norm_old;
    if (!x2 && c->norm_park_cap < park_words) {
      retire_or_free(c, c->d_norm_park);
      c->d_norm_park = nullptr;
      c->norm_park_cap = 0;
      HIPCHK(hipMalloc((void**)&c->d_norm_park, park_words * sizeof(double2)));
      c->norm_park_cap = park_words;
    }
    if (x2) {
      // both sub-transforms at once in one 1024-thread workgroup (norm_r16.h: r16x2), nothing parked
      constexpr size_t x2_lds = 2 * (size_t)hx::R16_LDS_DOUBLES * sizeof(double);
      bool attrx2 = false;   // (hxp::dyn_lds is idempotent per device)
      if (!attrx2) {
        HIPCHK(hxp::dyn_lds((const void*)hx::embed_norm_r16x2_kernel, (int)x2_lds));
        attrx2 = true;
      }
      HX_LAUNCH(hx::embed_norm_r16x2_kernel, dim3((unsigned)rows), dim3(2 * hx::R16_THREADS), x2_lds, ns, d_f, c->d_wtab,
                out2, direct);
    } else
    HX_LAUNCH(hx::embed_norm_quarter_split_kernel, dim3((unsigned)rows * (S / 2)), dim3(hx::NORM_THREADS),
                       16 * (size_t)H, ns, d_f, c->d_wtab, logn, logh, c->d_norm_park, c->d_norm2);
  } else {
    CHK(flush_xs(c));
    const int logh = std::min(logn, hx::NORM_MAX_LOGH);
    const unsigned H = 1u << logh, S = N >> logh;
    const unsigned threads = std::min<unsigned>(hx::NORM_THREADS, std::max<unsigned>(64u, H / 4));
    const size_t lds = std::max<size_t>(16 * (size_t)H, 256);
    HX_LAUNCH(hx::embed_norm_kernel, dim3((unsigned)rows * S), dim3(threads), lds, ns, d_f,
                       c->d_wtab, logn, logh, c->d_norm2);
  }
  HIPCHK(hipGetLastError());
  np.rows = rows;
  np.out = out_host;
  // the few words go to the pinned slot by a kernel writing host memory (hipHostMalloc memory is device-visible),
  // not by a device-to-host copy command: HX_NORM_MEMCPY=1 restores the copy (A/B)
  if (direct) {
    // (written by the norm kernel itself)
  } else if (by_copy) {
    HIPCHK(hipMemcpyAsync(np.pinned, c->d_norm2, sizeof(unsigned long long) * (size_t)rows,
                          hipMemcpyDeviceToHost, ns));
  } else {
    HX_LAUNCH(hx::copy_words_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ns,
              reinterpret_cast<uint64_t*>(np.pinned), reinterpret_cast<const uint64_t*>(c->d_norm2), (size_t)rows);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipEventRecord(np.ev, ns));
  c->norm_pending.push_back(np);
  slot_guard.armed = false;
  if (ns != c->stream) {
    HIPCHK(hipEventRecord(c->norm_out_ev, ns));
    (use_xs ? c->norm_reads_xs : c->norm_reads_frac) = true;
  }
  if (c->defer_norms)
    return HX_OK;
  return hx_norms_flush(c);
}

// Wait for a norm read-back: poll for up to 2 ms before handing the thread to hipEventSynchronize.  The host loop of a
// multiply waits for the previous multiply's norms while the current one runs; a sleeping wait wakes up on the host's
// timer granularity, and on part of this pool the CKKS loop then ran at exactly 3.000 ms per multiply with 2.90 ms of
// kernels in it (profiles/r03_bench_line_ckks65536_final.json: wall_us_per_multiply_of_the_batch).
static hipError_t wait_event(const hx_ctx* c, hipEvent_t ev)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    for (int i = 0; i < 64; i++) {
      const hipError_t e = hipEventQuery(ev);
      if (e != hipErrorNotReady)
        return e;
    }
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(c->sw.wait_poll_us))
      break;
  }
  return hipEventSynchronize(ev);
}

extern "C" int hx_norms_flush(hx_ctx* c)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null context");
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_norms_flush");
  for (auto& np : c->norm_pending) {
    HIPCHK(wait_event(c, np.ev));
    for (int r = 0; r < np.rows; r++) {
      double v;
      memcpy(&v, &np.pinned[r], 8);
      np.out[r] = sqrt(v);
    }
    c->norm_free.push_back(np);
  }
  c->norm_pending.clear();
  return HX_OK;
}

extern "C" int hx_ctx_defer_norms(hx_ctx* c, int on)
{
  if (!c)
    return fail(HX_ERR_INVALID, "null context");
  c->defer_norms = on != 0;
  if (!on)
    return hx_norms_flush(c);
  return HX_OK;
}

extern "C" int hx_embedding_norm(hx_ctx* c, const double* f_host, int rows, double* norms_out)
{
  if (!c || !f_host || !norms_out || rows < 1)
    return fail(HX_ERR_INVALID, "bad argument");
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_embedding_norm");
  const size_t n = (size_t)rows * c->phim;
  CHK(frac_begin(c, n));
  c->want_frac = false;
  HIPCHK(hipMemcpyAsync(c->d_frac, f_host, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  CHK(embed_norms(c, c->d_frac, rows, norms_out));
  return hx_norms_flush(c);  // host in, host out: always complete on return
}

// ------------------------------------------------------------------
// addPrimesAndScale / addPrimes / scaleDownToSet
// ------------------------------------------------------------------
static int scale_rows_by_primes(hx_poly* a, const int* add_idx, int nadd)
{
  hx_ctx* c = a->ctx;
  std::vector<uint64_t> f(a->nrows());
  for (int r = 0; r < a->nrows(); r++) {
    uint64_t q = c->primes[a->prime_idx[r]].q, v = 1;
    for (int i = 0; i < nadd; i++)
      v = hxh::mulmod(v, c->primes[add_idx[i]].q % q, q);
    f[r] = v;
  }
  return ew_scalar_rows<hx::EWS_MUL>(a, f.data());
}

extern "C" int hx_add_primes_and_scale(hx_poly* a, const int* add_idx, int nadd)
{
  if (!a)
    return fail(HX_ERR_INVALID, "null poly");
  if (nadd == 0)
    return HX_OK;
  hx_ctx* c = a->ctx;
  CTX_ENTER(c);
  CHK(check_rows(c, add_idx, nadd));
  for (int i = 0; i < nadd; i++)
    if (find_row(a->prime_idx, add_idx[i]) >= 0)
      return fail(HX_ERR_PRIMESET, "addPrimes can only be called on a disjoint set");
  int old = a->nrows();
  OWN(a);
  if (old > 0)
    CHK(scale_rows_by_primes(a, add_idx, nadd));
  CHK(poly_reserve(a, old + nadd));
  HIPCHK(hipMemsetAsync(a->d + (size_t)old * a->row_words(), 0,
                        (size_t)nadd * a->row_words() * 8, c->stream));
  for (int i = 0; i < nadd; i++)
    a->prime_idx.push_back(add_idx[i]);
  return HX_OK;
}

extern "C" int hx_add_primes(hx_poly* a, const int* add_idx, int nadd)
{
  if (!a)
    return fail(HX_ERR_INVALID, "null poly");
  if (nadd == 0)
    return HX_OK;
  hx_ctx* c = a->ctx;
  CTX_ENTER(c);
  CHK(check_rows(c, add_idx, nadd));
  for (int i = 0; i < nadd; i++)
    if (find_row(a->prime_idx, add_idx[i]) >= 0)
      return fail(HX_ERR_PRIMESET, "addPrimes can only be called on a disjoint set");
  int old = a->nrows();
  OWN(a);
  CHK(poly_reserve(a, old + nadd));
  size_t rw = a->row_words();
  if (old == 0) {  // special case for empty DCRT (src/DoubleCRT.cpp:579-585)
    HIPCHK(hipMemsetAsync(a->d, 0, (size_t)nadd * rw * 8, c->stream));
    a->prime_idx.assign(add_idx, add_idx + nadd);
    return HX_OK;
  }
  if (old > MAX_EXT_SRC)
    return fail(HX_ERR_UNSUPPORTED, "addPrimes from more than %d primes", MAX_EXT_SRC);
  // toPoly: inverse transform of a copy
  CHK(ensure_scratch(c, 0, (size_t)old * rw));
  CHK(dcopy(c, c->scratch[0], a->d, (size_t)old * rw));
  CHK(ntt_rows(c, c->scratch[0], a->prime_idx, old, 0, old, a->batch, true));
  std::vector<int> tgt(add_idx, add_idx + nadd);
  ExtPlan* pl;
  CHK(get_plan(c, a->prime_idx, tgt, 0, &pl));
  ExtArgs args;
  clear_args(args);
  args.src = c->scratch[0];
  args.dst = a->d;
  for (int k = 0; k < old; k++)
    args.src_row[k] = (uint16_t)k;
  for (int t = 0; t < nadd; t++)
    args.dst_row[t] = (uint16_t)(old + t);
  CHK(launch_extend(c, pl, args, rw));
  std::vector<int> all = a->prime_idx;
  for (int i = 0; i < nadd; i++)
    all.push_back(add_idx[i]);
  // FFT(poly, s1) on the new rows only
  if (old + nadd > MAX_ROWS)
    return fail(HX_ERR_UNSUPPORTED, "too many rows");
  CHK(ntt_rows(c, a->d, all, old + nadd, old, nadd, a->batch, false));
  a->prime_idx = all;
  return HX_OK;
}

// DoubleCRT::toPoly followed by PolyRed(.., t, abs = true) (src/DoubleCRT.cpp:925-1113,
// src/NumbTh.cpp:775-803): out[b][j] = (coefficient j of the centred polynomial) mod t in [0,t).
// This is the tail of SecKey::Decrypt (src/keys.cpp:1383-1405) without big integers: inverse
// transform of a copy, Garner mixed-radix digits, centring, residue modulo t -- exact.
extern "C" int hx_poly_rem(const hx_poly* a, uint64_t t, uint64_t* out_host)
{
  if (!a || !out_host)
    return fail(HX_ERR_INVALID, "null argument");
  if (t < 2 || t >= (1ull << 60))
    return fail(HX_ERR_INVALID, "modulus must be in [2, 2^60)");
  hx_ctx* c = a->ctx;
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_poly_rem");
  const int n = a->nrows();
  const size_t rw = a->row_words();
  if (n == 0) {  // the zero polynomial
    memset(out_host, 0, rw * 8);
    return HX_OK;
  }
  if (n > MAX_EXT_SRC)
    return fail(HX_ERR_UNSUPPORTED, "toPoly from more than %d primes on the device", MAX_EXT_SRC);
  CHK(ensure_scratch(c, 0, (size_t)(n + 1) * rw));
  CHK(dcopy(c, c->scratch[0], a->d, (size_t)n * rw));
  CHK(ntt_rows(c, c->scratch[0], a->prime_idx, n, 0, n, a->batch, true));
  std::vector<int> tgt(1, 0);
  std::vector<uint64_t> mod(1, t);
  ExtPlan* pl;
  CHK(get_plan(c, a->prime_idx, tgt, 0, &pl, &mod));
  ExtArgs args;
  clear_args(args);
  args.src = c->scratch[0];
  args.dst = c->scratch[0];
  for (int k = 0; k < n; k++)
    args.src_row[k] = (uint16_t)k;
  args.dst_row[0] = (uint16_t)n;
  CHK(launch_extend(c, pl, args, rw));
  HIPCHK(hipMemcpyAsync(out_host, c->scratch[0] + (size_t)n * rw, rw * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return HX_OK;
}

// others[0..nother): further polys with exactly a's prime rows and batch, mod-switched by the
// same launches when the fused single-prime path applies (returns HX_ERR_UNSUPPORTED otherwise
// when nother > 0, so that the caller can fall back to one call per poly).
// scaleDownToSet / bringToSet dropping SEVERAL primes, for all listed polys (one prime set, one
// batch) in one set of launches -- what every multiply after the first goes through (the special
// primes of the previous key switch, the small primes on the way out of a product).  Round 1 did this
// per polynomial and unfused (inverse transforms, basis extension writing delta, forward transforms,
// then a separate (c - delta)/P pass; the mod-up as a scaling pass of its own before).  Now:
//   1. per dropped prime, ONE launch of the mod-down prep kernel over all polys: inverse transform
//      of that row into the x block, the mod-up factor F = prod(added primes) riding on the last
//      stage's N^-1 twiddles (no scaled copy of the operand is ever made);
//   2. per poly, the basis-extension kernel on a plan whose tables carry P^-1 (get_plan scaled):
//      Garner + centring + ptxtSpace correction, out comes delta/P on every kept prime
//      (the dropped primes are taken in ascending order, so that the lazy Garner form applies to
//      any mix of prime sizes);
//   3. ONE launch of the apply kernel with a plain load over all polys: forward transform of
//      delta/P with the store  c_r <- c_r*(F/P) - NTT(.)  into a fresh, compact slab per poly --
//      out of place by construction, so lazily copied operands (hx_poly::Share) are never copied.
// Returns HX_ERR_UNSUPPORTED (nothing touched) when the shape is not covered; the callers then
// take the per-polynomial path.
// tsrc (tensorProduct folded in): the polys are the three product parts with reserved storage and the operands'
// prime list; their rows are formed from the operands' rows inside the transforms and the results go into the
// parts' own slabs.
static int scale_down_multi_fused(hx_poly** ps, int np, const std::vector<int>& drop_in,
                                  const std::vector<int>& keep, uint64_t ptxt, const int* add_idx, int nadd,
                                  const hx::TensorSrc* tsrc = nullptr)
{
  hx_poly* a = ps[0];
  hx_ctx* c = a->ctx;
  const int nd = (int)drop_in.size(), nk = (int)keep.size();
  if (!c->pow2 || c->logn < 13 || c->logn > 15 || nd < 2 || nd > 64 || nk > MAX_ROWS || np > hx::MD_MAXPOLY)
    return HX_ERR_UNSUPPORTED;
  for (int i = 0; i < np; i++)
    if (!ps[i]->owns)
      return HX_ERR_UNSUPPORTED;  // caller-owned storage wants its result in place
  if (tsrc && (np != 3 || c->sw.no_tensor_multi))
    return HX_ERR_UNSUPPORTED;
  const size_t rw = a->row_words();
  const int batch = a->batch;
  std::vector<int> drop = drop_in;
  std::sort(drop.begin(), drop.end(), [&](int x, int y) { return c->primes[x].q < c->primes[y].q; });
  // kept row t: prime keep[t]; its c_r lives in row in_row[t] of the input slab (-1: a row the
  // fused mod-up adds, c_r = 0)
  std::vector<int> in_row(nk);
  for (int t = 0; t < nk; t++)
    in_row[t] = find_row(a->prime_idx, keep[t]);  // (the added primes are not among the poly's rows)
  ExtPlan* pl;
  CHK(get_plan(c, drop, keep, ptxt > 1 ? ptxt : 0, &pl, nullptr, /*scaled=*/true));
  CHK(ensure_scratch(c, 0, (size_t)np * nd * rw));  // x blocks  [poly][dropped prime][batch][N]
  CHK(ensure_scratch(c, 1, (size_t)np * nk * rw));  // delta / P [poly][kept row][batch][N]
  // per-row constants of the apply launch, cached per (dropped set, kept rows, added primes)
  std::vector<uint64_t> key;
  key.push_back(0xD1D1D1D1ull);
  for (int d : drop)
    key.push_back((uint64_t)d);
  key.push_back(0xFFFFull << 32);
  for (int j = 0; j < nadd; j++)
    key.push_back(0xA000000ull | (uint64_t)add_idx[j]);
  NttRows kr;
  std::vector<hx::ModDownRow> hr(nk);
  for (int t = 0; t < nk; t++) {
    const int pr = keep[t];
    const uint64_t q = c->primes[pr].q;
    kr.row[t] = (uint16_t)(in_row[t] < 0 ? 0 : in_row[t]);
    kr.prime[t] = (uint16_t)pr;
    uint64_t P = 1;
    for (int d : drop)
      P = hxh::mulmod(P, c->primes[d].q % q, q);
    uint64_t cf = hxh::invmod(P, q);
    if (cf == 0)
      return fail(HX_ERR_INVALID, "dropped primes are not invertible modulo a kept prime");
    memset(&hr[t], 0, sizeof hr[t]);
    hr[t].out_row = (uint32_t)t;
    if (in_row[t] < 0) {
      hr[t].mode = 2;  // c_r = 0: cf stays 0
    } else {
      for (int j = 0; j < nadd; j++)
        cf = hxh::mulmod(cf, c->primes[add_idx[j]].q % q, q);
      hr[t].mode = nadd > 0 ? 1 : 0;
      hr[t].cf.w = cf;
      hr[t].cf.wp = tw_companion(c->primes[pr], cf);
      hr[t].cf_r2 = tw_r2(c->primes[pr], cf);
    }
    key.push_back(((uint64_t)pr << 28) | ((uint64_t)kr.row[t] << 12) | hr[t].mode);
  }
  auto it = c->plans.find(key);
  if (it == c->plans.end()) {
    ExtPlan* rp = new ExtPlan();
    memset(&rp->dev, 0, sizeof rp->dev);
    HIPCHK(hipMalloc(&rp->blob, sizeof(hx::ModDownRow) * nk));
    HIPCHK(hipMemcpy(rp->blob, hr.data(), sizeof(hx::ModDownRow) * nk, hipMemcpyHostToDevice));
    it = c->plans.emplace(key, rp).first;
  }
  // output slabs first (see the single-prime path: nothing is released before everything is taken)
  const int ncap = nk + 2;
  const size_t nbytes = (size_t)ncap * rw * 8;
  uint64_t* fresh[hx::MD_MAXPOLY] = {nullptr};
  auto drop_fresh = [&]() {
    for (int i = 0; i < np; i++)
      if (fresh[i])
        pool_free(c, fresh[i], nbytes);
  };
  PolyBases pb, pbo;
  memset(&pb, 0, sizeof pb);
  memset(&pbo, 0, sizeof pbo);
  pb.n = pbo.n = np;
  for (int i = 0; i < np; i++) {
    if (tsrc) {  // nothing of the part exists yet: its own slab takes the result
      if (ps[i]->cap_rows < nk)
        return HX_ERR_UNSUPPORTED;
      pb.d[i] = pbo.d[i] = ps[i]->d;
      continue;
    }
    hipError_t pe = pool_alloc(c, nbytes, (void**)&fresh[i]);
    if (pe != hipSuccess) {
      drop_fresh();
      return fail(HX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", nbytes, hipGetErrorString(pe));
    }
    pb.d[i] = ps[i]->d;
    pbo.d[i] = fresh[i];
  }
  int rc = HX_OK;
  // 1. inverse transforms of the dropped rows: all polys and up to MD_MAXDROP dropped primes per launch
  for (int j0 = 0; j0 < nd && rc == HX_OK; j0 += hx::MD_MAXDROP) {
    const int nj = std::min(hx::MD_MAXDROP, nd - j0);
    hx::PrepMulti M;
    memset(&M, 0, sizeof M);
    ModDownPrep P;
    memset(&P, 0, sizeof P);
    P.xs = c->scratch[0] + (size_t)j0 * rw;
    P.poly_stride = (uint64_t)nd * rw;
    P.has_up = nadd > 0 ? 1 : 0;
    for (int j = 0; j < nj; j++) {
      const int dprime = drop[j0 + j];
      const uint64_t qd = c->primes[dprime].q;
      M.row[j] = (uint16_t)find_row(a->prime_idx, dprime);
      M.prime[j] = (uint16_t)dprime;
      if (nadd > 0) {
        uint64_t F = 1;
        for (int i = 0; i < nadd; i++)
          F = hxh::mulmod(F, c->primes[add_idx[i]].q % qd, qd);
        M.up[2 * j].w = hxh::mulmod(F, c->primes[dprime].last_s, qd);
        M.up[2 * j].wp = tw_companion(c->primes[dprime], M.up[2 * j].w);
        M.up[2 * j + 1].w = hxh::mulmod(F, c->primes[dprime].last_n, qd);
        M.up[2 * j + 1].wp = tw_companion(c->primes[dprime], M.up[2 * j + 1].w);
      }
    }
    hipError_t e = tsrc ? hx::launch_moddown_prep_multi_tensor_pow2(c->logn, *tsrc, M, nj, batch, P, c->d_primes, c->d_tw,
                                                                    c->stream)
                        : hx::launch_moddown_prep_multi_pow2(c->logn, pb, M, nj, batch, P, c->d_primes, c->d_tw, c->stream);
    if (e != hipSuccess)
      rc = fail(HX_ERR_DEVICE, "mod-down launch failed: %s", hipGetErrorString(e));
  }
  // 2. delta / P on the kept primes, per poly
  for (int i = 0; i < np && rc == HX_OK; i++) {
    ExtArgs args;
    clear_args(args);
    args.src = c->scratch[0] + (size_t)i * nd * rw;
    args.dst = c->scratch[1] + (size_t)i * nk * rw;
    for (int k = 0; k < nd; k++)
      args.src_row[k] = (uint16_t)k;
    for (int t = 0; t < nk; t++)
      args.dst_row[t] = (uint16_t)t;
    // scratch[1] is read by step 3 only, whose row loads are declared with LOAD_BOUND 8 -- 4 on rows of Proth-form
    // primes, which is what the fast kernel's Proth-form target sums deliver (as digits_lazy_ok)
    args.lazy_out = pl->dev.lazy_tight_ok ? 1 : 0;   // (per plan: every Proth-form target on its Montgomery branch)
    if (c->want_frac) {
      args.frac = frac_take(c, rw);
      if (!args.frac)
        rc = fail(HX_ERR_INVALID, "internal: fraction buffer too small");
    }
    if (rc == HX_OK)
      rc = launch_extend(c, pl, args, rw);
  }
  // 3. forward transforms with the subtraction and the division in the store
  if (rc == HX_OK) {
    ModDownApply A;
    memset(&A, 0, sizeof A);
    A.rows = reinterpret_cast<const hx::ModDownRow*>(it->second->blob);
    A.delta = c->scratch[1];
    A.delta_poly_stride = (uint64_t)nk * rw;
    hipError_t e = tsrc ? hx::launch_moddown_apply_plain_tensor_pow2(c->logn, *tsrc, pbo, kr, nk, batch, A, c->d_primes,
                                                                     c->d_tw, c->stream)
                        : hx::launch_moddown_apply_plain_pow2(c->logn, pb, pbo, kr, nk, batch, A, c->d_primes, c->d_tw,
                                                              c->stream);
    if (e != hipSuccess)
      rc = fail(HX_ERR_DEVICE, "mod-down launch failed: %s", hipGetErrorString(e));
  }
  if (rc != HX_OK) {
    drop_fresh();
    return rc;
  }
  for (int i = 0; i < np; i++) {
    if (!tsrc) {
      storage_release(ps[i]);  // (a slab shared with a lazy copy stays with its other holders)
      ps[i]->d = fresh[i];
      ps[i]->cap_rows = ncap;
    }
    ps[i]->prime_idx = keep;
  }
  return HX_OK;
}

static int scale_down_impl(hx_poly* a, hx_poly** others, int nother, const int* drop_idx, int ndrop,
                           uint64_t ptxt, const int* add_idx = nullptr, int nadd = 0,
                           const hx::TensorSrc* tsrc = nullptr)
{
  // tsrc (tensorProduct folded in): a / others are the three product parts, storage reserved, prime list =
  // the operands', contents not yet computed -- only the fused single-prime path can take them
  // add_idx (fused bringToSet): the polys are first mod-switched UP by these primes
  // (addPrimesAndScale); only the fused single-prime path folds that in, otherwise the caller
  // gets HX_ERR_UNSUPPORTED and does the two steps separately.

  if (!a)
    return fail(HX_ERR_INVALID, "null poly");
  hx_ctx* c = a->ctx;
  CTX_ENTER(c);
  if (c->want_frac)
    CHK(flush_xs(c));  // the scratch slots of an earlier fused block are about to be reused
  if (ptxt < 1)
    return fail(HX_ERR_INVALID, "ptxtSpace must be at least 1");
  // the fused bringToSet appends the mod-up's primes to the row list before its fallible steps (slab
  // and scratch allocation, the invertibility check): an error return takes them off again, so the
  // poly never advertises rows that were not written
  struct PrimeRollback {
    hx_poly* a;
    int n = 0;
    ~PrimeRollback()
    {
      if (n > 0)
        a->prime_idx.resize(a->prime_idx.size() - (size_t)n);
    }
  } rollback{a};
  if (nadd > 0 && ndrop != 1) {
    // several dropped primes behind a mod-up: the batched path or nothing (the caller then does
    // addPrimesAndScale and the mod-down as two steps)
    std::vector<int> drop, keep;
    for (int r = 0; r < a->nrows(); r++) {
      bool d = false;
      for (int i = 0; i < ndrop; i++)
        d = d || drop_idx[i] == a->prime_idx[r];
      (d ? drop : keep).push_back(a->prime_idx[r]);
    }
    for (int i = 0; i < nadd; i++) {
      if (find_row(a->prime_idx, add_idx[i]) >= 0)
        return HX_ERR_UNSUPPORTED;
      for (int j = 0; j < ndrop; j++)
        if (drop_idx[j] == add_idx[i])
          return HX_ERR_UNSUPPORTED;
      keep.push_back(add_idx[i]);
    }
    if (drop.size() < 2 || (int)keep.size() == nadd)
      return HX_ERR_UNSUPPORTED;
    hx_poly* ps[hx::MD_MAXPOLY];
    if (1 + nother > hx::MD_MAXPOLY)
      return HX_ERR_UNSUPPORTED;
    ps[0] = a;
    for (int i = 0; i < nother; i++)
      ps[1 + i] = others[i];
    return scale_down_multi_fused(ps, 1 + nother, drop, keep, ptxt, add_idx, nadd, tsrc);
  }
  if (nadd > 0) {
    bool ok = ndrop == 1 && c->pow2 && c->logn >= 13 && c->logn <= 15 && find_row(a->prime_idx, drop_idx[0]) >= 0 &&
              a->nrows() + nadd - 1 <= MAX_ROWS && ptxt < ((uint64_t)1 << 62);
    for (int i = 0; i < nadd && ok; i++)
      if (find_row(a->prime_idx, add_idx[i]) >= 0 || add_idx[i] == drop_idx[0])
        ok = false;
    // the fused kernels take |S| <= ptxtSpace/2 + 1 as already reduced modulo every kept prime
    for (int i = 0; i < nadd && ok; i++)
      ok = ptxt / 2 + 2 < c->primes[add_idx[i]].q;
    for (int r = 0; r < a->nrows() && ok; r++)
      ok = a->prime_idx[r] == drop_idx[0] || ptxt / 2 + 2 < c->primes[a->prime_idx[r]].q;
    if (!ok)
      return HX_ERR_UNSUPPORTED;
    // make room and append the new (never read) rows; a shared poly (copy-on-write) gets its output
    // slab below instead
    if (!a->shared())
      CHK(poly_reserve(a, a->nrows() + nadd));
    for (int i = 0; i < nother; i++)
      if (!others[i]->shared())
        CHK(poly_reserve(others[i], a->nrows() + nadd));
    for (int i = 0; i < nadd; i++)
      a->prime_idx.push_back(add_idx[i]);
    rollback.n = nadd;
  }
  const int nrows_old = a->nrows() - nadd;
  // diff = getIndexSet() / s : only primes actually present are dropped
  std::vector<int> drop, keep;
  for (int r = 0; r < a->nrows(); r++) {
    bool d = false;
    for (int i = 0; i < ndrop; i++)
      if (drop_idx[i] == a->prime_idx[r])
        d = true;
    (d ? drop : keep).push_back(a->prime_idx[r]);
  }
  if (drop.empty()) {  // nothing to do; a requested norm is that of delta = 0
    if (tsrc)
      return HX_ERR_UNSUPPORTED;   // (none of the listed primes is there: the caller forms the product on its own)
    if (c->want_frac) {
      double* fr = frac_take(c, a->row_words() * (size_t)(1 + nother));
      if (fr)
        HIPCHK(hipMemsetAsync(fr, 0, a->row_words() * (size_t)(1 + nother) * sizeof(double), c->stream));
    }
    return HX_OK;
  }
  if (keep.empty())
    return fail(HX_ERR_PRIMESET, "s and the index set must have some intersection");
  if ((int)drop.size() > MAX_EXT_SRC)
    return fail(HX_ERR_UNSUPPORTED, "scaleDownToSet dropping more than %d primes", MAX_EXT_SRC);
  size_t rw = a->row_words();
  int nd = (int)drop.size(), nk = (int)keep.size();
  bool small_S = true;  // |S| <= ptxtSpace/2 + 1 is a reduced residue of every kept prime
  for (int pr : keep)
    small_S = small_S && ptxt / 2 + 2 < c->primes[pr].q;
  if (tsrc && nd >= 2) {   // (nadd == 0 here: the mod-up case went to the batched path above)
    hx_poly* ps[hx::MD_MAXPOLY];
    if (1 + nother > hx::MD_MAXPOLY)
      return HX_ERR_UNSUPPORTED;
    ps[0] = a;
    for (int i = 0; i < nother; i++)
      ps[1 + i] = others[i];
    return scale_down_multi_fused(ps, 1 + nother, drop, keep, ptxt, nullptr, 0, tsrc);
  }
  if (tsrc && !(nd == 1 && nk <= MAX_ROWS && small_S))
    return HX_ERR_UNSUPPORTED;   // (checked before anything was changed: nadd > 0 passed its own test above)
  if (nd == 1 && c->pow2 && c->logn >= 13 && c->logn <= 15 && nk <= MAX_ROWS && ptxt < ((uint64_t)1 << 62) && small_S) {
    // fused path: [inverse NTT of the dropped row + delta preparation] then [forward NTT of
    // delta on every kept row, with  c <- (c - delta) / qd  in its store].  The last row takes
    // the dropped row's slot, so no compaction copy is needed.
    const int dprime = drop[0], drow = find_row(a->prime_idx, dprime), last = a->nrows() - 1;
    const uint64_t qd = c->primes[dprime].q;
    // inputs and outputs: in place, except for polys that share their slab with a copy
    // (hx_poly_copy is lazy) -- those read the shared slab and write a fresh one, which is what
    // makes `Ctxt tmp = other; tmp.bringToSet(s)` cost no copy at all
    PolyBases pb, pbo;
    memset(&pb, 0, sizeof pb);
    memset(&pbo, 0, sizeof pbo);
    pb.n = pbo.n = 1 + nother;
    hx_poly* ps[hx::MD_MAXPOLY];
    uint64_t* fresh[hx::MD_MAXPOLY] = {nullptr};
    ps[0] = a;
    for (int i = 0; i < nother; i++)
      ps[1 + i] = others[i];
    const int ncap = a->nrows() + 2;
    const size_t nbytes = (size_t)ncap * rw * 8;
    for (int i = 0; i < pb.n; i++)
      if (ps[i]->share && !ps[i]->shared())
        CHK(poly_own(ps[i]));  // last holder of its slab: plain in-place
    // (all output slabs are taken BEFORE any shared slab is released: two listed polys may share
    // one slab, and a slab returned to the pool could come straight back as somebody's output)
    auto drop_fresh = [&]() {
      for (int i = 0; i < pb.n; i++)
        if (fresh[i])
          pool_free(c, fresh[i], nbytes);
    };
    for (int i = 0; i < pb.n; i++) {
      pb.d[i] = pbo.d[i] = ps[i]->d;
      if (ps[i]->shared()) {
        hipError_t pe = pool_alloc(c, nbytes, (void**)&fresh[i]);
        if (pe != hipSuccess) {
          drop_fresh();
          return fail(HX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", nbytes, hipGetErrorString(pe));
        }
        pbo.d[i] = fresh[i];
      }
    }
    struct FreshGuard {  // error paths below give the output slabs back
      decltype(drop_fresh)& f;
      bool armed = true;
      ~FreshGuard()
      {
        if (armed)
          f();
      }
    } fresh_guard{drop_fresh};
    CHK(ensure_scratch(c, 0, rw * pb.n));
    CHK(ensure_scratch(c, 1, rw * pb.n));
    hx::ModDownPrep P;
    memset(&P, 0, sizeof P);
    P.xs = c->scratch[0];
    P.S = reinterpret_cast<int64_t*>(c->scratch[1]);
    P.half = (qd - 1) / 2;
    if (ptxt > 1) {
      P.ptxt = ptxt;
      P.ptxt_mu64 = (uint64_t)((((hxh::u128)1) << 64) / ptxt);
      int kb = hxh::bitlen(ptxt);
      P.ptxt_k = (uint32_t)kb;
      P.ptxt_mu = (uint64_t)((((hxh::u128)1) << (2 * kb)) / ptxt);
      P.qd_mod_p = qd % ptxt;
      P.qdinv_mod_p = hxh::invmod(qd % ptxt, ptxt);
      if (P.qdinv_mod_p == 0)
        return fail(HX_ERR_INVALID, "dropped primes are not invertible modulo ptxtSpace");
    }
    P.qd = qd;
    if (nadd > 0) {
      uint64_t F = 1;
      for (int i = 0; i < nadd; i++)
        F = hxh::mulmod(F, c->primes[add_idx[i]].q % qd, qd);
      P.has_up = 1;
      P.upS.w = hxh::mulmod(F, c->primes[dprime].last_s, qd);
      P.upS.wp = tw_companion(c->primes[dprime], P.upS.w);
      P.upN.w = hxh::mulmod(F, c->primes[dprime].last_n, qd);
      P.upN.wp = tw_companion(c->primes[dprime], P.upN.w);
    }
    // per-row constants (cached per (dropped prime, kept rows and their output slots))
    std::vector<uint64_t> key;
    key.push_back(0xD0D0D0D0ull);
    key.push_back((uint64_t)dprime);
    for (int j = 0; j < nadd; j++)
      key.push_back(0xA000000ull | (uint64_t)add_idx[j]);
    NttRows kr;
    std::vector<hx::ModDownRow> hr(nk);
    int i = 0;
    for (int r = 0; r <= last; r++) {
      if (r == drow)
        continue;
      const int pr = a->prime_idx[r];
      const uint64_t q = c->primes[pr].q;
      kr.row[i] = (uint16_t)r;
      kr.prime[i] = (uint16_t)pr;
      uint64_t qdm = qd % q, inv = hxh::invmod(qdm, q);
      hr[i].qdm.w = qdm;
      hr[i].qdm.wp = hxh::shoup(qdm, q);
      hr[i].inv.w = inv;
      hr[i].inv.wp = tw_companion(c->primes[pr], inv);
      hr[i].out_row = (uint32_t)((r == last && drow != last) ? drow : r);
      hr[i].mode = 0;
      hr[i].cf = hr[i].inv;
      hr[i].cf_r2 = tw_r2(c->primes[pr], inv);
      if (nadd > 0) {
        if (r >= nrows_old) {
          hr[i].mode = 2;  // a row the mod-up adds: c_r = 0, i.e. cf = 0 (its slot is never initialised)
          hr[i].cf.w = hr[i].cf.wp = 0;
          hr[i].cf_r2 = 0;
        } else {
          uint64_t F = 1;
          for (int j = 0; j < nadd; j++)
            F = hxh::mulmod(F, c->primes[add_idx[j]].q % q, q);
          uint64_t cf = hxh::mulmod(F, inv, q);
          hr[i].mode = 1;
          hr[i].cf.w = cf;
          hr[i].cf.wp = tw_companion(c->primes[pr], cf);
          hr[i].cf_r2 = tw_r2(c->primes[pr], cf);
        }
      }
      key.push_back(((uint64_t)pr << 28) | ((uint64_t)hr[i].out_row << 12) | hr[i].mode);
      i++;
    }
    auto it = c->plans.find(key);
    if (it == c->plans.end()) {
      ExtPlan* pl = new ExtPlan();
      memset(&pl->dev, 0, sizeof pl->dev);
      HIPCHK(hipMalloc(&pl->blob, sizeof(hx::ModDownRow) * nk));
      HIPCHK(hipMemcpy(pl->blob, hr.data(), sizeof(hx::ModDownRow) * nk, hipMemcpyHostToDevice));
      it = c->plans.emplace(key, pl).first;
    }
    hx::ModDownApply A;
    A.xs = c->scratch[0];
    A.S = reinterpret_cast<const int64_t*>(c->scratch[1]);
    A.rows = reinterpret_cast<const hx::ModDownRow*>(it->second->blob);
    hipError_t e = tsrc ? hx::launch_moddown_tensor_pow2(c->logn, *tsrc, pbo, drow, dprime, kr, nk, a->batch, P, A,
                                                         c->d_primes, c->d_tw, c->stream)
                        : hx::launch_moddown_pow2(c->logn, pb, pbo, drow, dprime, kr, nk, a->batch, P, A,
                                                  c->d_primes, c->d_tw, c->stream);
    if (e != hipSuccess)
      return fail(HX_ERR_DEVICE, "mod-down launch failed: %s", hipGetErrorString(e));
    fresh_guard.armed = false;
    for (int i = 0; i < pb.n; i++)
      if (fresh[i]) {  // the shared input slab stays with its other holders (or goes back to the
        storage_release(ps[i]);  // pool, whose reuse is ordered behind the kernels just enqueued)
        ps[i]->d = fresh[i];
        ps[i]->cap_rows = ncap;
      }
    if (c->want_frac) {
      const size_t n = rw * (size_t)pb.n;
      double* fr = frac_take(c, n);
      if (!fr)
        return fail(HX_ERR_INVALID, "internal: fraction buffer too small");
      if (c->xs_rows == 0 && c->frac_pos == n && !c->want_fdelta) {
        // the whole norm batch comes from this call: the norm kernel reads (x, S) itself
        c->xs_first = 0;
        c->xs_rows = pb.n * a->batch;
        c->xs_inv_qd = 1.0 / (double)qd;
      } else {
        HX_LAUNCH(hx::frac_from_xs_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)),
                           dim3(256), 0, c->stream, c->scratch[0],
                           reinterpret_cast<const int64_t*>(c->scratch[1]), 1.0 / (double)qd, fr, n);
        HIPCHK(hipGetLastError());
      }
    }
    rollback.n = 0;
    if (drow != last)
      a->prime_idx[drow] = a->prime_idx[last];
    a->prime_idx.pop_back();
    for (int i = 0; i < nother; i++)
      others[i]->prime_idx = a->prime_idx;
    return HX_OK;
  }
  if (nd >= 2 && 1 + nother <= hx::MD_MAXPOLY) {
    hx_poly* ps[hx::MD_MAXPOLY];
    ps[0] = a;
    for (int i = 0; i < nother; i++)
      ps[1 + i] = others[i];
    int rc = scale_down_multi_fused(ps, 1 + nother, drop, keep, ptxt, nullptr, 0);
    if (rc != HX_ERR_UNSUPPORTED)
      return rc;
  }
  if (nother > 0)
    return HX_ERR_UNSUPPORTED;
  // toPoly(delta, diff): inverse transform of the dropped rows, out of place into scratch rows
  // of the same index (no row copies)
  CHK(ensure_scratch(c, 0, (size_t)a->nrows() * rw));
  std::vector<std::pair<int, int>> drows;
  for (int k = 0; k < nd; k++)
    drows.emplace_back(find_row(a->prime_idx, drop[k]), drop[k]);
  CHK(ntt_list(c, a->d, c->scratch[0], drows, a->batch, true));
  ExtPlan* pl;
  CHK(get_plan(c, drop, keep, ptxt > 1 ? ptxt : 0, &pl));
  // delta on the kept primes goes into a fresh slab that then becomes the polynomial's storage:
  // new[w] = (old[row of keep[w]] - delta[w]) / diffProd, so neither removePrimes' compaction nor
  // a second buffer is needed
  uint64_t* nd_buf = nullptr;
  const int ncap = nk + 2;
  const size_t nbytes = (size_t)ncap * rw * 8;
  hipError_t pe = pool_alloc(c, nbytes, (void**)&nd_buf);
  if (pe != hipSuccess)
    return fail(HX_ERR_NOMEM, "hipMalloc(%zu) failed: %s", nbytes, hipGetErrorString(pe));
  ExtArgs args;
  clear_args(args);
  args.src = c->scratch[0];
  args.dst = nd_buf;
  for (int k = 0; k < nd; k++)
    args.src_row[k] = (uint16_t)drows[k].first;
  for (int t = 0; t < nk; t++)
    args.dst_row[t] = (uint16_t)t;
  int rc = HX_OK;
  if (c->want_frac) {
    args.frac = frac_take(c, rw);
    if (!args.frac)
      rc = fail(HX_ERR_INVALID, "internal: fraction buffer too small");
  }
  if (rc == HX_OK)
    rc = launch_extend(c, pl, args, rw);
  if (rc == HX_OK)
    rc = ntt_rows(c, nd_buf, keep, nk, 0, nk, a->batch, false);
  if (rc != HX_OK) {
    pool_free(c, nd_buf, nbytes);
    return rc;
  }
  // *this -= delta; *this /= diffProd, reading the kept rows where they are
  RowMap2 map;
  RowScalars sc;
  for (int r = 0; r < nk; r++) {
    uint64_t q = c->primes[keep[r]].q, v = 1;
    for (int k = 0; k < nd; k++)
      v = hxh::mulmod(v, c->primes[drop[k]].q % q, q);
    uint64_t inv = hxh::invmod(v, q);
    map.p[r] = (uint16_t)keep[r];
    map.brow[r] = (uint16_t)find_row(a->prime_idx, keep[r]);
    sc.c[r] = inv;
    sc.cp[r] = hxh::shoup(inv, q);
  }
  HX_LAUNCH(hx::sub_scale_from_kernel, ew_grid(rw, nk), dim3(256), 0, c->stream, nd_buf,
                     a->d, map, sc, rw, c->d_primes);
  HIPCHK(hipGetLastError());
  if (a->owns) {
    storage_release(a);  // stream-ordered reuse (a shared slab stays with its other holders)
    a->d = nd_buf;
    a->cap_rows = ncap;
  } else {  // caller-owned storage: the compact result goes back into it
    CHK(dcopy(c, a->d, nd_buf, (size_t)nk * rw));
    pool_free(c, nd_buf, nbytes);
  }
  a->prime_idx = keep;
  return HX_OK;
}

extern "C" int hx_scale_down(hx_poly* a, const int* drop_idx, int ndrop, uint64_t ptxt)
{
  return scale_down_impl(a, nullptr, 0, drop_idx, ndrop, ptxt);
}

// The same mod-switch applied to several DoubleCRT objects that share one prime set and batch
// (the parts of the ciphertexts being brought to a common level, src/Ctxt.cpp:462-465): one pair
// of launches covers all of them when a single prime is dropped.
extern "C" int hx_scale_down_multi(hx_poly** polys, int npoly, const int* drop_idx, int ndrop,
                                   uint64_t ptxt)
{
  if (!polys || npoly < 1)
    return fail(HX_ERR_INVALID, "bad argument");
  bool same = npoly <= hx::MD_MAXPOLY;
  for (int i = 0; i < npoly; i++) {
    if (!polys[i])
      return fail(HX_ERR_INVALID, "null poly");
    if (polys[i]->ctx != polys[0]->ctx || polys[i]->batch != polys[0]->batch ||
        polys[i]->prime_idx != polys[0]->prime_idx)
      same = false;
    for (int j = 0; j < i; j++)
      if (polys[j] == polys[i])
        return fail(HX_ERR_INVALID, "the same poly listed twice");
  }
  if (same && npoly > 1) {
    int rc = scale_down_impl(polys[0], polys + 1, npoly - 1, drop_idx, ndrop, ptxt);
    if (rc != HX_ERR_UNSUPPORTED)
      return rc;
  }
  for (int i = 0; i < npoly; i++)
    CHK(scale_down_impl(polys[i], nullptr, 0, drop_idx, ndrop, ptxt));
  return HX_OK;
}

// Ctxt::bringToSet on several parts (src/Ctxt.cpp:373-389): modUpToSet by add_idx
// (addPrimesAndScale) immediately followed by modDownToSet dropping drop_idx.  When exactly one
// prime is dropped the up-scaling is folded into the fused mod-down kernels (the scaled rows
// and the zero rows are never materialised); otherwise the two steps run one after the other.
extern "C" int hx_bring_to_set_multi(hx_poly** polys, int npoly, const int* add_idx, int nadd,
                                     const int* drop_idx, int ndrop, uint64_t ptxt)
{
  if (!polys || npoly < 1 || (nadd > 0 && !add_idx) || (ndrop > 0 && !drop_idx))
    return fail(HX_ERR_INVALID, "bad argument");
  bool same = npoly <= hx::MD_MAXPOLY;
  for (int i = 0; i < npoly; i++) {
    if (!polys[i])
      return fail(HX_ERR_INVALID, "null poly");
    if (polys[i]->ctx != polys[0]->ctx || polys[i]->batch != polys[0]->batch ||
        polys[i]->prime_idx != polys[0]->prime_idx)
      same = false;
    for (int j = 0; j < i; j++)
      if (polys[j] == polys[i])
        return fail(HX_ERR_INVALID, "the same poly listed twice");
  }
  CHK(check_rows(polys[0]->ctx, add_idx, nadd));
  if (same && nadd > 0 && ndrop >= 1) {
    int rc = scale_down_impl(polys[0], polys + 1, npoly - 1, drop_idx, ndrop, ptxt, add_idx, nadd);
    if (rc != HX_ERR_UNSUPPORTED)
      return rc;
  }
  for (int i = 0; i < npoly && nadd > 0; i++)
    CHK(hx_add_primes_and_scale(polys[i], add_idx, nadd));
  if (ndrop == 0)
    return HX_OK;
  return hx_scale_down_multi(polys, npoly, drop_idx, ndrop, ptxt);
}

// ---- measured mod-switch noise (src/Ctxt.cpp:466-530): the same operations, additionally
// returning embeddingLargestCoeff(fdelta) per (poly, batch element), fdelta = delta / diffProd.
static int finish_norms(hx_ctx* c, int rc, size_t doubles, int rows, double* norms, double* frac_host)
{
  c->want_frac = false;
  if (rc != HX_OK)
    return rc;
  // blocks no kernel wrote (nothing to drop) stay zero from frac_begin
  if (frac_host)
    HIPCHK(hipMemcpyAsync(frac_host, c->d_frac, doubles * sizeof(double), hipMemcpyDeviceToHost,
                          c->stream));
  return embed_norms(c, c->d_frac, rows, norms);
}

extern "C" int hx_scale_down_multi_norms(hx_poly** polys, int npoly, const int* drop_idx, int ndrop,
                                         uint64_t ptxt, double* norms, double* fdelta)
{
  if (!polys || npoly < 1 || !polys[0] || !norms)
    return fail(HX_ERR_INVALID, "bad argument");
  hx_ctx* c = polys[0]->ctx;
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_scale_down_multi_norms (a norm read-back)");
  const size_t rw = polys[0]->row_words();
  const int batch = polys[0]->batch;
  CHK(frac_begin(c, (size_t)npoly * rw));
  c->want_fdelta = fdelta != nullptr;
  int rc = hx_scale_down_multi(polys, npoly, drop_idx, ndrop, ptxt);
  c->want_fdelta = false;
  return finish_norms(c, rc, (size_t)npoly * rw, npoly * batch, norms, fdelta);
}

extern "C" int hx_bring_to_set_multi_norms(hx_poly** polys, int npoly, const int* add_idx, int nadd,
                                           const int* drop_idx, int ndrop, uint64_t ptxt,
                                           double* norms)
{
  if (!polys || npoly < 1 || !polys[0] || !norms)
    return fail(HX_ERR_INVALID, "bad argument");
  hx_ctx* c = polys[0]->ctx;
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_bring_to_set_multi_norms (a norm read-back)");
  const size_t rw = polys[0]->row_words();
  const int batch = polys[0]->batch;
  CHK(frac_begin(c, (size_t)npoly * rw));
  int rc = hx_bring_to_set_multi(polys, npoly, add_idx, nadd, drop_idx, ndrop, ptxt);
  return finish_norms(c, rc, (size_t)npoly * rw, npoly * batch, norms, nullptr);
}

static int tensor_launch(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1, uint64_t* o0,
                         uint64_t* o1, uint64_t* o2, const uint64_t* scale_per_row);
// Ctxt::tensorProduct of two canonical ciphertexts (src/Ctxt.cpp:1563-1608) immediately followed by
// Ctxt::bringToSet of the product (reLinearize's dropSmallAndSpecialPrimes, :720-760): o0, o1, o2 = the three
// product parts (1), (s), (s^2) mod-switched up by add_idx and down by drop_idx.  With one dropped prime the
// product parts are never materialised on the operands' prime set (TensorSrc); any other shape runs the two
// steps one after the other -- same result either way.
static int tensor_bring_to_set(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1,
                               hx_poly* o0, hx_poly* o1, hx_poly* o2, const int* add_idx, int nadd,
                               const int* drop_idx, int ndrop, uint64_t ptxt)
{
  hx_ctx* c = c0->ctx;
  hx_poly* os[3] = {o0, o1, o2};
  const hx_poly* in[4] = {c0, c1, d0, d1};
  for (auto* p : in)
    if (p->ctx != c || p->batch != c0->batch || p->prime_idx != c0->prime_idx)
      return fail(HX_ERR_PRIMESET, "tensorProduct: parts must be defined relative to the same set of primes");
  for (int i = 0; i < 3; i++) {
    if (os[i]->ctx != c || os[i]->batch != c0->batch)
      return fail(HX_ERR_INVALID, "Context mismatch");
    for (auto* p : in)
      if (p == os[i])
        return fail(HX_ERR_INVALID, "hx_tensor_bring_to_set: an output aliases an operand");
    for (int j = 0; j < i; j++)
      if (os[j] == os[i])
        return fail(HX_ERR_INVALID, "the same poly listed twice");
  }
  CHK(check_rows(c, add_idx, nadd));
  for (auto* o : os) {
    OWN(o);
    CHK(poly_reserve(o, c0->nrows() + nadd, /*keep=*/false));
    o->prime_idx = c0->prime_idx;
  }
  if (c0->nrows() <= MAX_ROWS && ndrop >= 1) {
    hx::TensorSrc T{c0->d, c1->d, d0->d, d1->d};
    int rc = scale_down_impl(o0, os + 1, 2, drop_idx, ndrop, ptxt, add_idx, nadd, &T);
    if (rc != HX_ERR_UNSUPPORTED)
      return rc;
    for (auto* o : os)
      o->prime_idx = c0->prime_idx;   // (untouched by an unsupported attempt; restated for clarity)
  }
  CHK(tensor_launch(c0, c1, d0, d1, o0->d, o1->d, o2->d, nullptr));
  return hx_bring_to_set_multi(os, 3, add_idx, nadd, drop_idx, ndrop, ptxt);
}
extern "C" int hx_tensor_bring_to_set(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1,
                                      hx_poly* o0, hx_poly* o1, hx_poly* o2, const int* add_idx, int nadd,
                                      const int* drop_idx, int ndrop, uint64_t ptxt)
{
  if (!c0 || !c1 || !d0 || !d1 || !o0 || !o1 || !o2 || (nadd > 0 && !add_idx) || (ndrop > 0 && !drop_idx))
    return fail(HX_ERR_INVALID, "null argument");
  CTX_ENTER(c0->ctx);
  return tensor_bring_to_set(c0, c1, d0, d1, o0, o1, o2, add_idx, nadd, drop_idx, ndrop, ptxt);
}
extern "C" int hx_tensor_bring_to_set_norms(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1,
                                            hx_poly* o0, hx_poly* o1, hx_poly* o2, const int* add_idx, int nadd,
                                            const int* drop_idx, int ndrop, uint64_t ptxt, double* norms)
{
  if (!c0 || !c1 || !d0 || !d1 || !o0 || !o1 || !o2 || !norms || (nadd > 0 && !add_idx) || (ndrop > 0 && !drop_idx))
    return fail(HX_ERR_INVALID, "null argument");
  hx_ctx* c = c0->ctx;
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_tensor_bring_to_set_norms (a norm read-back)");
  const size_t rw = c0->row_words();
  CHK(frac_begin(c, 3 * rw));
  int rc = tensor_bring_to_set(c0, c1, d0, d1, o0, o1, o2, add_idx, nadd, drop_idx, ndrop, ptxt);
  return finish_norms(c, rc, 3 * rw, 3 * c0->batch, norms, nullptr);
}

// ------------------------------------------------------------------
// breakIntoDigits
// ------------------------------------------------------------------
// coef: inverse-transformed copy of the ctxt rows (modified in place),
// dig : [ndig][nall][batch][N] output in the coefficient domain.
static int break_digits_coef(hx_ctx* c, uint64_t* coef, const std::vector<int>& own,
                             const int* dig_idx, const int* dig_off, int ndig,
                             const std::vector<int>& all, uint64_t* dig, size_t rw,
                             bool copy_own = true, std::vector<int>* owner_out = nullptr)
{
  int nall = (int)all.size();
  // which digit owns each ctxt prime
  std::vector<int> owner(all.size(), -1);
  for (int d = 0; d < ndig; d++)
    for (int p = dig_off[d]; p < dig_off[d + 1]; p++) {
      int pos = find_row(all, dig_idx[p]);
      if (pos < 0 || find_row(own, dig_idx[p]) < 0)
        return fail(HX_ERR_PRIMESET, "digit prime %d is not a row of the operand", dig_idx[p]);
      if (owner[pos] >= 0)
        return fail(HX_ERR_INVALID, "digit sets overlap");
      owner[pos] = d;
    }
  for (size_t r = 0; r < own.size(); r++)
    if (owner[find_row(all, own[r])] < 0)
      return fail(HX_ERR_PRIMESET, "prime %d of the operand is in no digit", own[r]);
  for (int d = 0; d < ndig; d++) {
    std::vector<int> src(dig_idx + dig_off[d], dig_idx + dig_off[d + 1]);
    if (src.empty())
      return fail(HX_ERR_INVALID, "empty digit");
    // targets: first the rows owned by later digits (they are also updated in place), then
    // the rest (earlier digits' rows and the special primes)
    std::vector<int> tgt;
    for (int r = 0; r < nall; r++)
      if (owner[r] > d)
        tgt.push_back(all[r]);
    int nu = (int)tgt.size();
    for (int r = 0; r < nall; r++)
      if (owner[r] < d)
        tgt.push_back(all[r]);
    ExtPlan* pl;
    CHK(get_plan(c, src, tgt, 0, &pl));
    ExtArgs args;
    clear_args(args);
    args.src = coef;
    args.dst = dig;
    args.upd = coef;
    for (size_t k = 0; k < src.size(); k++) {
      args.src_row[k] = (uint16_t)find_row(own, src[k]);
      if (copy_own)
        args.own_dst_row[k] = (uint16_t)(d * nall + find_row(all, src[k]));
    }
    for (size_t t = 0; t < tgt.size(); t++) {
      int pos = find_row(all, tgt[t]);
      args.dst_row[t] = (uint16_t)(d * nall + pos);
      if (owner[pos] > d)
        args.upd_row[t] = (uint16_t)find_row(own, tgt[t]);
    }
    args.nu = nu;
    if (c->want_frac) {
      args.frac = frac_take(c, rw);
      if (!args.frac)
        return fail(HX_ERR_INVALID, "internal: fraction buffer too small");
    }
    CHK(launch_extend(c, pl, args, rw));
  }
  if (owner_out)
    *owner_out = owner;
  return HX_OK;
}

// Fused variant for hx_mul_relin: digits are contiguous row runs of `own` (checked), own rows
// are not materialised.  Returns HX_ERR_UNSUPPORTED when the shape does not fit (caller falls
// back to break_digits_coef).
// lazy_out: the extension words may be left in [0,6q) (fast kernels only; the caller's forward transform takes lazy input)
static int break_digits_fused(hx_ctx* c, const uint64_t* coef, const std::vector<int>& own,
                              const int* dig_idx, const int* dig_off, int ndig,
                              const std::vector<int>& all, uint64_t* dig, size_t rw,
                              std::vector<int>* owner_out, bool lazy_out = false)
{
  const int L = (int)own.size(), nall = (int)all.size();
  if (ndig > hx::KS_MAXD || L > 64 || dig_off[ndig] != L)
    return HX_ERR_UNSUPPORTED;
  for (int r = 0; r < L; r++)
    if (all[r] != own[r] || dig_idx[r] != own[r])
      return HX_ERR_UNSUPPORTED;  // digits must tile the operand's rows in order
  int nmax = 0;
  hx::BreakArgs A;
  memset(&A, 0, sizeof A);
  A.src = coef;
  A.dst = dig;
  A.L = L;
  A.nall = nall;
  A.ndig = ndig;
  std::vector<int> owner(nall, -1);
  for (int d = 0; d < ndig; d++) {
    A.off[d] = dig_off[d];
    int n = dig_off[d + 1] - dig_off[d];
    if (n < 1 || n > 16)
      return HX_ERR_UNSUPPORTED;
    nmax = std::max(nmax, n);
    std::vector<int> src(dig_idx + dig_off[d], dig_idx + dig_off[d + 1]), tgt;
    for (int r = 0; r < nall; r++) {
      if (r >= dig_off[d] && r < dig_off[d + 1])
        owner[r] = d;
      else
        tgt.push_back(all[r]);
    }
    ExtPlan* pl;
    CHK(get_plan(c, src, tgt, 0, &pl));
    A.plan[d] = pl->dev;
  }
  A.off[ndig] = dig_off[ndig];
  if (c->want_frac) {
    A.frac = frac_take(c, (size_t)ndig * rw);
    if (!A.frac)
      return fail(HX_ERR_INVALID, "internal: fraction buffer too small");
  }
  dim3 grid((unsigned)((rw + hx::BRK_THREADS - 1) / hx::BRK_THREADS)), block(hx::BRK_THREADS);
  size_t lds = (size_t)L * hx::BRK_THREADS * 8;
  bool fast = true;
  for (int d = 0; d < ndig; d++)
    fast = fast && A.plan[d].fast_ok;
  A.redo = nullptr;
  for (int d = 0; d < ndig; d++)   // (lazy words only from plans whose Proth-form targets all take the Montgomery branch)
    lazy_out = lazy_out && A.plan[d].lazy_tight_ok;
  if (fast) {
    const int lds_rows = std::max(1, L - hx::break_fast_n0(A));
    // (the column caps the resident waves at 160 KiB / (rows x 8 B x 64 lanes) per CU: 16 rows at the CKKS chain = five
    // waves per SIMD where the registers allow seven -- worth 24 % by the HX_BRK_LDS_PAD probe; the two ways tried to
    // move the column out of the LDS did not pay: profiles/r06_ab_digit_kernel_occupancy.json)
    for (const void* f : {(const void*)hx::break_digits_fast_kernel<true, false>, (const void*)hx::break_digits_fast_kernel<false, false>,
                          (const void*)hx::break_digits_fast_kernel<true, true>, (const void*)hx::break_digits_fast_kernel<false, true>})
      HIPCHK(hxp::dyn_lds(f, 64 * hx::BRK_THREADS * 8));
    // (HX_BRK_LDS_PAD: unused rows on top, to measure what the kernel's occupancy is worth -- profiles/r06_ab_digit_kernel_occupancy)
    const size_t lds_fast = (size_t)(lds_rows + std::max(0, std::min(40, c->sw.brk_lds_pad_rows))) * hx::BRK_THREADS * 8;
    bool hps = rw < ((size_t)1 << 32);
    for (int d = 0; d < ndig; d++)
      hps = hps && A.plan[d].hps_ok && (int)A.plan[d].n >= c->sw.brk_hps_min_n;
#define HX_BRK_LAUNCH(H, Z, G) HX_LAUNCH((hx::break_digits_fast_kernel<H, Z>), G, block, lds_fast, c->stream, A, rw)
    if (hps) {   // HPS form, then Garner over the coefficients it could not vouch for (rns_kernels.h: ExtRep)
      CHK(redo_prepare(c, rw, &A.redo));
      if (lazy_out) {
        HX_BRK_LAUNCH(true, true, grid);
        HX_BRK_LAUNCH(false, true, REDO_GRID);
      } else {
        HX_BRK_LAUNCH(true, false, grid);
        HX_BRK_LAUNCH(false, false, REDO_GRID);
      }
    } else if (lazy_out) {
      HX_BRK_LAUNCH(false, true, grid);
    } else {
      HX_BRK_LAUNCH(false, false, grid);
    }
#undef HX_BRK_LAUNCH
  } else if (nmax <= 8) {
    bool attr8 = false;   // (hxp::dyn_lds is idempotent per device)
    if (!attr8) {
      HIPCHK(hxp::dyn_lds((const void*)hx::break_digits_kernel<8>, 64 * hx::BRK_THREADS * 8));
      attr8 = true;
    }
    HX_LAUNCH((hx::break_digits_kernel<8>), grid, block, lds, c->stream, A, rw);
  } else {
    bool attr16 = false;   // (hxp::dyn_lds is idempotent per device)
    if (!attr16) {
      HIPCHK(hxp::dyn_lds((const void*)hx::break_digits_kernel<16>, 64 * hx::BRK_THREADS * 8));
      attr16 = true;
    }
    HX_LAUNCH((hx::break_digits_kernel<16>), grid, block, lds, c->stream, A, rw);
  }
  HIPCHK(hipGetLastError());
  if (owner_out)
    *owner_out = owner;
  return HX_OK;
}

static int build_all(const hx_poly* a, const int* sp_idx, int nsp, std::vector<int>& all)
{
  all = a->prime_idx;
  for (int i = 0; i < nsp; i++) {
    if (find_row(all, sp_idx[i]) >= 0)
      return fail(HX_ERR_PRIMESET,
                  "Special primes and CtxtPart's index set have non-empty intersection");
    all.push_back(sp_idx[i]);
  }
  return HX_OK;
}

extern "C" int hx_break_into_digits(const hx_poly* a, const int* dig_idx, const int* dig_off,
                                    int ndig, const int* sp_idx, int nsp, hx_poly* out)
{
  if (!a || !out || !dig_idx || !dig_off || ndig < 1)
    return fail(HX_ERR_INVALID, "bad argument");
  hx_ctx* c = a->ctx;
  if (out->ctx != c || out->batch != a->batch)
    return fail(HX_ERR_INVALID, "Context mismatch");
  CTX_ENTER(c);
  CHK(check_rows(c, sp_idx, nsp));
  std::vector<int> all;
  CHK(build_all(a, sp_idx, nsp, all));
  int nall = (int)all.size(), L = a->nrows();
  if (nall > MAX_ROWS)
    return fail(HX_ERR_UNSUPPORTED, "too many rows");
  size_t rw = a->row_words();
  OWN(out);
  CHK(poly_reserve(out, ndig * nall, /*keep=*/out == a));
  out->prime_idx.clear();
  for (int d = 0; d < ndig; d++)
    out->prime_idx.insert(out->prime_idx.end(), all.begin(), all.end());
  CHK(ensure_scratch(c, 0, (size_t)L * rw));
  CHK(dcopy(c, c->scratch[0], a->d, (size_t)L * rw));
  CHK(ntt_rows(c, c->scratch[0], a->prime_idx, L, 0, L, a->batch, true));
  CHK(break_digits_coef(c, c->scratch[0], a->prime_idx, dig_idx, dig_off, ndig, all, out->d, rw));
  CHK(ntt_rows(c, out->d, all, nall, 0, ndig * nall, a->batch, false));
  return HX_OK;
}

// DoubleCRT::breakIntoDigits with its return value (src/DoubleCRT.cpp:538-545): norms[d*batch+b] =
// embeddingLargestCoeff(digit_d of element b) / P_d  (P_d = product of the digit's primes; the
// host multiplies it back in extended range, digits reach 2^2000).
extern "C" int hx_break_into_digits_norms(const hx_poly* a, const int* dig_idx, const int* dig_off,
                                          int ndig, const int* sp_idx, int nsp, hx_poly* out,
                                          double* norms)
{
  if (!a || !norms || ndig < 1)
    return fail(HX_ERR_INVALID, "bad argument");
  hx_ctx* c = a->ctx;
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_break_into_digits_norms (a norm read-back)");
  const size_t rw = a->row_words();
  CHK(frac_begin(c, (size_t)ndig * rw));
  int rc = hx_break_into_digits(a, dig_idx, dig_off, ndig, sp_idx, nsp, out);
  return finish_norms(c, rc, (size_t)ndig * rw, ndig * a->batch, norms, nullptr);
}

// ------------------------------------------------------------------
// key-switch matrices, tensor, key switch, multiply
// ------------------------------------------------------------------
extern "C" int hx_ksk_create(hx_ctx* c, int ndig, const int* row_idx, int nrows, const uint64_t* b,
                             const uint64_t* a, hx_ksk** out)
{
  if (!c || !out || !b || !a || ndig < 1)
    return fail(HX_ERR_INVALID, "bad argument");
  CTX_ENTER(c);
  CHK(check_rows(c, row_idx, nrows));
  hx_ksk* k = new hx_ksk();
  struct Guard {   // a failing allocation / copy below returns through HIPCHK
    hx_ksk* k;
    ~Guard()
    {
      if (k) {
        hipFree(k->d_b);
        hipFree(k->d_a);
        delete k;
      }
    }
  } guard{k};
  k->ctx = c;
  k->ndig = ndig;
  k->row_idx.assign(row_idx, row_idx + nrows);
  size_t bytes = (size_t)ndig * nrows * c->phim * 8;
  HIPCHK(hipMalloc((void**)&k->d_b, bytes));
  HIPCHK(hipMalloc((void**)&k->d_a, bytes));
  HIPCHK(hipMemcpy(k->d_b, b, bytes, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(k->d_a, a, bytes, hipMemcpyHostToDevice));
  c->refs++;
  guard.k = nullptr;
  *out = k;
  return HX_OK;
}
extern "C" int hx_ksk_shape(const hx_ksk* k, int* ndig, int* nrows, int* row_idx_out)
{
  if (!k || !ndig || !nrows)
    return fail(HX_ERR_INVALID, "null argument");
  *ndig = k->ndig;
  *nrows = (int)k->row_idx.size();
  if (row_idx_out)
    for (size_t r = 0; r < k->row_idx.size(); r++)
      row_idx_out[r] = k->row_idx[r];
  return HX_OK;
}
extern "C" int hx_ksk_download(const hx_ksk* k, uint64_t* b, uint64_t* a)
{
  if (!k || !b || !a)
    return fail(HX_ERR_INVALID, "null argument");
  hx_ctx* c = k->ctx;
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_ksk_download (a synchronous copy)");
  const size_t bytes = (size_t)k->ndig * k->row_idx.size() * c->phim * 8;
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(b, k->d_b, bytes, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(a, k->d_a, bytes, hipMemcpyDeviceToHost));
  return HX_OK;
}
extern "C" int hx_ksk_destroy(hx_ksk* k)
{
  if (!k)
    return HX_OK;
  hipSetDevice(k->ctx->device);
  hipStreamSynchronize(k->ctx->stream);
  hipFree(k->d_b);
  hipFree(k->d_a);
  ctx_release(k->ctx);
  delete k;
  return HX_OK;
}

static int tensor_launch(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0,
                         const hx_poly* d1, uint64_t* o0, uint64_t* o1, uint64_t* o2,
                         const uint64_t* scale_per_row)
{
  hx_ctx* c = c0->ctx;
  int rows = c0->nrows();
  if (rows > MAX_ROWS)
    return fail(HX_ERR_UNSUPPORTED, "too many rows");
  const hx_poly* ps[3] = {c1, d0, d1};
  for (auto* p : ps)
    if (p->ctx != c || p->batch != c0->batch || p->prime_idx != c0->prime_idx)
      return fail(HX_ERR_PRIMESET,
                  "tensorProduct: parts must be defined relative to the same set of primes");
  RowMap map;
  RowScalars sc;
  memset(&sc, 0, sizeof sc);
  for (int r = 0; r < rows; r++) {
    map.p[r] = (uint16_t)c0->prime_idx[r];
    if (scale_per_row) {
      uint64_t q = c->primes[c0->prime_idx[r]].q;
      sc.c[r] = scale_per_row[r] % q;
      sc.cp[r] = hxh::shoup(sc.c[r], q);
    }
  }
  size_t rw = c0->row_words();
  HX_LAUNCH(hx::tensor_kernel, ew_grid(rw, rows), dim3(256), 0, c->stream, c0->d, c1->d,
                     d0->d, d1->d, o0, o1, o2, map, sc, scale_per_row ? 1 : 0, rw, c->d_primes);
  HIPCHK(hipGetLastError());
  return HX_OK;
}

extern "C" int hx_tensor(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0,
                         const hx_poly* d1, hx_poly* o0, hx_poly* o1, hx_poly* o2)
{
  if (!c0 || !c1 || !d0 || !d1 || !o0 || !o1 || !o2)
    return fail(HX_ERR_INVALID, "null poly");
  CTX_ENTER(c0->ctx);
  hx_poly* os[3] = {o0, o1, o2};
  for (auto* o : os) {
    if (o->ctx != c0->ctx || o->batch != c0->batch)
      return fail(HX_ERR_INVALID, "Context mismatch");
    OWN(o);
    CHK(poly_reserve(o, c0->nrows(), /*keep=*/false));
    o->prime_idx = c0->prime_idx;
  }
  return tensor_launch(c0, c1, d0, d1, o0->d, o1->d, o2->d, nullptr);
}

// own_src/owner: when given, digit `owner[r]` of row r is not read from `dig` but rebuilt in the
// evaluation domain from the s^2 part and the earlier digits (src/DoubleCRT.cpp:552-556 applied
// row-wise): own = (...((c - d_0)/P_0 - d_1)/P_1 ...).
static int keyswitch_launch(hx_ctx* c, const uint64_t* dig, const hx_ksk* W,
                            const std::vector<int>& all, int batch, uint64_t* out0, uint64_t* out1,
                            int accumulate_rows, const uint64_t* own_src = nullptr,
                            const std::vector<int>* owner = nullptr,
                            const std::vector<std::vector<int>>* digit_primes = nullptr,
                            int ndig = -1, const uint64_t* t0s = nullptr, const uint64_t* t1s = nullptr,
                            const hx::TensorSrc* ts = nullptr)
{
  int nall = (int)all.size();
  if (ndig < 0)
    ndig = W->ndig;
  if (nall > MAX_ROWS)
    return fail(HX_ERR_UNSUPPORTED, "too many rows");
  RowMap2 map;  // p = prime of the row, brow = its row inside W (W may cover more primes)
  for (int r = 0; r < nall; r++) {
    int wr = find_row(W->row_idx, all[r]);
    if (wr < 0)
      return fail(HX_ERR_PRIMESET, "key-switching matrix is not defined on the operand's primes");
    map.p[r] = (uint16_t)all[r];
    map.brow[r] = (uint16_t)wr;
  }
  size_t rw = (size_t)batch * c->phim;
  int lazy = W->ndig <= 8 ? 1 : 0;
  for (int r : all)
    if (hxh::bitlen(c->primes[r].q) > 60)
      lazy = 0;
  const hx::KsFix* d_fix = nullptr;
  if (own_src || ts) {
    // cached table: owner digit of every row and P_e^-1 mod q_row for the earlier digits
    std::vector<uint64_t> key;
    key.push_back(0xF1F1F1F1ull);
    for (int r : all)
      key.push_back((uint64_t)r);
    for (auto& dp : *digit_primes) {
      key.push_back(0xFFFFull);
      for (int p : dp)
        key.push_back((uint64_t)p);
    }
    auto it = c->plans.find(key);
    if (it == c->plans.end()) {
      if ((int)digit_primes->size() > hx::KS_MAXD)
        return fail(HX_ERR_UNSUPPORTED, "more than %d digits", hx::KS_MAXD);
      std::vector<hx::KsFix> h(nall);
      for (int r = 0; r < nall; r++) {
        memset(&h[r], 0, sizeof(hx::KsFix));
        h[r].owner = (*owner)[r];
        uint64_t q = c->primes[all[r]].q;
        for (int e = 0; e < (int)digit_primes->size(); e++) {
          uint64_t pe = 1;
          for (int p : (*digit_primes)[e])
            pe = hxh::mulmod(pe, c->primes[p].q % q, q);
          uint64_t inv = (h[r].owner >= 0 && e < h[r].owner) ? hxh::invmod(pe, q) : 0;
          h[r].pinv[e].w = inv;
          h[r].pinv[e].wp = hxh::shoup(inv, q);
        }
        uint64_t ps = 1;
        for (int t = 0; t < nall; t++)
          if ((*owner)[t] < 0)
            ps = hxh::mulmod(ps, c->primes[all[t]].q % q, q);
        h[r].pscale.w = ps;
        h[r].pscale.wp = hxh::shoup(ps, q);
      }
      ExtPlan* pl = new ExtPlan();
      memset(&pl->dev, 0, sizeof pl->dev);
      HIPCHK(hipMalloc(&pl->blob, sizeof(hx::KsFix) * nall));
      HIPCHK(hipMemcpy(pl->blob, h.data(), sizeof(hx::KsFix) * nall, hipMemcpyHostToDevice));
      it = c->plans.emplace(key, pl).first;
    }
    d_fix = reinterpret_cast<const hx::KsFix*>(it->second->blob);
  }
#define HX_KS_LAUNCH(ND)                                                                                           \
  HX_LAUNCH(hx::keyswitch_kernel<ND>, ew_grid(rw, nall), dim3(256), 0, c->stream, dig, W->d_b, W->d_a, out0, out1,  \
            map, ndig, nall, (int)W->row_idx.size(), batch, c->phim, accumulate_rows, c->d_primes, own_src, d_fix,  \
            lazy, d_fix ? t0s : nullptr, d_fix ? t1s : nullptr,                                                    \
            ts ? *ts : hx::TensorSrc{nullptr, nullptr, nullptr, nullptr})
  switch (ndig) {
    case 2: HX_KS_LAUNCH(2); break;
    case 3: HX_KS_LAUNCH(3); break;
    case 4: HX_KS_LAUNCH(4); break;
    default: HX_KS_LAUNCH(0); break;
  }
#undef HX_KS_LAUNCH
  HIPCHK(hipGetLastError());
  return HX_OK;
}

extern "C" int hx_key_switch_digits(const hx_poly* digits, const hx_ksk* W, hx_poly* out0,
                                    hx_poly* out1)
{
  if (!digits || !W || !out0 || !out1)
    return fail(HX_ERR_INVALID, "null argument");
  hx_ctx* c = digits->ctx;
  CTX_ENTER(c);
  // out0/out1 live on ctxt primes (possibly a lower level than W was made for) followed by the
  // special primes; the operand then has the first ndig <= W->ndig digits (src/DoubleCRT.cpp:485-493
  // keeps the digits that still intersect the prime set, which are the leading ones)
  const int nall = out0->nrows();
  if (nall == 0 || out1->prime_idx != out0->prime_idx || out0->batch != digits->batch ||
      out1->batch != digits->batch)
    return fail(HX_ERR_PRIMESET, "Ctxt::addPart: ctxt has primes not in part");
  if (digits->nrows() % nall != 0 || digits->nrows() / nall > W->ndig || digits->nrows() == 0)
    return fail(HX_ERR_INVALID, "W must have as many columns as there are digits");
  const int ndig = digits->nrows() / nall;
  for (int d = 0; d < ndig; d++)
    for (int r = 0; r < nall; r++)
      if (digits->prime_idx[(size_t)d * nall + r] != out0->prime_idx[r])
        return fail(HX_ERR_PRIMESET, "digit rows do not match the ciphertext's primes");
  OWN(out0);
  OWN(out1);
  return keyswitch_launch(c, digits->d, W, out0->prime_idx, digits->batch, out0->d, out1->d, nall,
                          nullptr, nullptr, nullptr, ndig);
}

// Ctxt::keySwitchPart on the s^2 part (src/Ctxt.cpp:805-842): t2e = its evaluation rows on `own`
// (left untouched), all = own followed by the special primes; accumulates into out0/out1 whose
// first L rows already hold the scaled parts (1), (s).
// t0s/t1s (optional): the unscaled parts (1),(s); the key-switch kernel then multiplies them by
// the special primes itself and out0/out1 need not be initialised.
static int relin_core(hx_ctx* c, const uint64_t* t2e, const std::vector<int>& own,
                      const std::vector<int>& all, const hx_ksk* W, const int* dig_idx,
                      const int* dig_off, int ndig, int batch, uint64_t* out0, uint64_t* out1,
                      const uint64_t* t0s = nullptr, const uint64_t* t1s = nullptr, const hx::TensorSrc* ts = nullptr)
{
  // ts (hx_mul_relin): the tensor product is folded in -- t2e is null; the s^2 part's coefficient rows come from
  // the inverse transform of a1 * b1 formed on load, and the key-switch kernel forms the parts (1), (s) and the
  // s^2 evaluation rows it needs from the four operand parts (own[r] is row r of each of them)
  const int L = (int)own.size(), nall = (int)all.size();
  const size_t rw = (size_t)batch * c->phim;
  // scratch: [2] = s^2 part in the coefficient domain; [1] = digits (ndig*nall rows; a digit's
  // own rows are never materialised)
  CHK(ensure_scratch(c, 2, (size_t)L * rw));
  if (ts) {
    NttRows nr;
    for (int r = 0; r < L; r++) {
      nr.row[r] = (uint16_t)r;
      nr.prime[r] = (uint16_t)own[r];
    }
    hipError_t e = hx::launch_ntt_inv_mul_pow2(c->logn, ts->a1, ts->b1, c->scratch[2], nr, L, batch, c->d_primes,
                                               c->d_tw, c->stream);
    if (e != hipSuccess)
      return fail(HX_ERR_DEVICE, "NTT launch failed: %s", hipGetErrorString(e));
  } else {
    std::vector<std::pair<int, int>> rows;
    for (int r = 0; r < L; r++)
      rows.emplace_back(r, own[r]);
    CHK(ntt_list(c, t2e, c->scratch[2], rows, batch, true));  // toPoly side, out of place
  }
  // (after the inverse transform is enqueued: taking slot 1 makes the stream wait for a norm kernel that
  // still reads the preceding mod-switch's S there -- it runs next to the inverse transform meanwhile)
  CHK(ensure_scratch(c, 1, (size_t)ndig * nall * rw));
  std::vector<int> owner;
  {
    // the extension words go straight into the forward transform below: where that takes lazy input they are left
    // unreduced (three conditional subtractions less per word)
    int rc = break_digits_fused(c, c->scratch[2], own, dig_idx, dig_off, ndig, all, c->scratch[1],
                                rw, &owner, digits_lazy_ok(c));
    if (rc == HX_ERR_UNSUPPORTED)
      rc = break_digits_coef(c, c->scratch[2], own, dig_idx, dig_off, ndig, all, c->scratch[1], rw,
                             /*copy_own=*/false, &owner);
    if (rc != HX_OK)
      return rc;
  }
  // forward NTT of the extension rows only: D*(L+K) - L transforms, as in the reference
  uint64_t* digits_eval = c->scratch[1];
  {
    std::vector<std::pair<int, int>> rows;
    for (int d = 0; d < ndig; d++)
      for (int r = 0; r < nall; r++)
        if (owner[r] != d)
          rows.emplace_back(d * nall + r, all[r]);
    // N = 2^15: out of place, so that the two-workgroups-per-row forward kernel applies (ntt_launch); the key switch
    // reads the extension rows only, all of which are written there
    if (c->pow2 && c->logn == 15 && c->sw.half15) {
      CHK(ensure_scratch(c, 10, (size_t)ndig * nall * rw));
      digits_eval = c->scratch[10];
    }
    CHK(ntt_list(c, c->scratch[1], digits_eval, rows, batch, false, /*lazy_in=*/true));
  }
  std::vector<std::vector<int>> dprimes(ndig);
  for (int d = 0; d < ndig; d++)
    dprimes[d].assign(dig_idx + dig_off[d], dig_idx + dig_off[d + 1]);
  return keyswitch_launch(c, digits_eval, W, all, batch, out0, out1, L, t2e, &owner, &dprimes, ndig,
                          t0s, t1s, ts);
}

static std::vector<uint64_t> special_factor(hx_ctx* c, const std::vector<int>& own, const int* sp,
                                            int nsp)
{
  std::vector<uint64_t> f(own.size());
  for (size_t r = 0; r < own.size(); r++) {
    uint64_t q = c->primes[own[r]].q, v = 1;
    for (int s = 0; s < nsp; s++)
      v = hxh::mulmod(v, c->primes[sp[s]].q % q, q);
    f[r] = v;
  }
  return f;
}

extern "C" int hx_mul_relin(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0,
                            const hx_poly* d1, const hx_ksk* W, const int* dig_idx,
                            const int* dig_off, int ndig, hx_poly* out0, hx_poly* out1)
{
  if (!c0 || !c1 || !d0 || !d1 || !W || !out0 || !out1 || !dig_idx || !dig_off)
    return fail(HX_ERR_INVALID, "null argument");
  hx_ctx* c = c0->ctx;
  CTX_ENTER(c);
  if (ndig > W->ndig || ndig < 1)
    return fail(HX_ERR_INVALID, "W must have as many columns as there are digits");
  int L = c0->nrows(), nall = (int)W->row_idx.size(), K = nall - L;
  if (K < 0)
    return fail(HX_ERR_PRIMESET, "key-switching matrix has fewer primes than the ciphertext");
  for (int r = 0; r < L; r++)
    if (W->row_idx[r] != c0->prime_idx[r])
      return fail(HX_ERR_PRIMESET, "key-switching matrix rows must start with the ctxt primes");
  if (out0->batch != c0->batch || out1->batch != c0->batch || out0->ctx != c || out1->ctx != c)
    return fail(HX_ERR_INVALID, "Context mismatch");
  {   // every operand check comes before the outputs are touched: a failing call leaves them as they were
    const hx_poly* ops4[4] = {c0, c1, d0, d1};
    for (auto* p : ops4)
      if (p->ctx != c || p->batch != c0->batch || p->prime_idx != c0->prime_idx)
        return fail(HX_ERR_PRIMESET, "tensorProduct: parts must be defined relative to the same set of primes");
  }
  size_t rw = c0->row_words();
  // (an output that aliases an input keeps its rows when it has to grow)
  OWN(out0);
  OWN(out1);
  CHK(poly_reserve(out0, nall, out0 == c0 || out0 == c1 || out0 == d0 || out0 == d1));
  CHK(poly_reserve(out1, nall, out1 == c0 || out1 == c1 || out1 == d0 || out1 == d1));
  out0->prime_idx = W->row_idx;
  out1->prime_idx = W->row_idx;
  // the tensor product folded into the inverse transform's load and the key-switch kernel (no tensor_kernel pass,
  // the three product parts are never written): power-of-two rings the row kernels take, outputs that are not
  // operands (the key-switch kernel reads operand rows of the coefficient it writes)
  const bool fuse_off = c->sw.no_mulrelin_fuse;
  const hx_poly* ins[4] = {c0, c1, d0, d1};
  bool alias = false;
  for (auto* p : ins)
    alias = alias || p == out0 || p == out1 || p->d == out0->d || p->d == out1->d;
  if (!fuse_off && !alias && c->pow2 && c->logn >= 13 && c->logn <= 15 && L <= MAX_ROWS) {
    hx::TensorSrc T{c0->d, c1->d, d0->d, d1->d};
    return relin_core(c, nullptr, c0->prime_idx, W->row_idx, W, dig_idx, dig_off, ndig, c0->batch, out0->d, out1->d,
                      nullptr, nullptr, &T);
  }
  CHK(ensure_scratch(c, 0, (size_t)L * rw));  // s^2 part, evaluation domain
  // tensorProduct + (parts 1,s) addPrimesAndScale(special)
  std::vector<uint64_t> f = special_factor(c, c0->prime_idx, W->row_idx.data() + L, K);
  CHK(tensor_launch(c0, c1, d0, d1, out0->d, out1->d, c->scratch[0], f.data()));
  return relin_core(c, c->scratch[0], c0->prime_idx, W->row_idx, W, dig_idx, dig_off, ndig,
                    c0->batch, out0->d, out1->d);
}

// hx_mul_relin with the digit norms Ctxt::keySwitchPart feeds into the noise estimate (as hx_relinearize_norms)
extern "C" int hx_mul_relin_norms(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1,
                                  const hx_ksk* W, const int* dig_idx, const int* dig_off, int ndig, hx_poly* out0,
                                  hx_poly* out1, double* norms)
{
  if (!c0 || !norms || ndig < 1)
    return fail(HX_ERR_INVALID, "bad argument");
  hx_ctx* c = c0->ctx;
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_mul_relin_norms (a norm read-back)");
  const size_t rw = c0->row_words();
  CHK(frac_begin(c, (size_t)ndig * rw));
  int rc = hx_mul_relin(c0, c1, d0, d1, W, dig_idx, dig_off, ndig, out0, out1);
  return finish_norms(c, rc, (size_t)ndig * rw, ndig * c0->batch, norms, nullptr);
}

// Ctxt::reLinearize for a 3-part ciphertext (1, s, s^2) (src/Ctxt.cpp:720-786): parts 1 and s get
// addPrimesAndScale(special), part s^2 goes through keySwitchPart.  W may cover more ctxt primes
// than the ciphertext currently has (lower level); digits are the context digits restricted to
// the ciphertext's primes.  out0/out1 end up on primes(t*) followed by sp_idx.
extern "C" int hx_relinearize(const hx_poly* t0, const hx_poly* t1, const hx_poly* t2,
                              const hx_ksk* W, const int* dig_idx, const int* dig_off, int ndig,
                              const int* sp_idx, int nsp, hx_poly* out0, hx_poly* out1)
{
  // t1 == NULL: no part pointing at s -- the (1, s(X^k)) ciphertext Ctxt::smartAutomorph
  // relinearises after Ctxt::automorph (src/Ctxt.cpp:2437-2515); t2 is then the s(X^k) part
  if (!t0 || !t2 || !W || !out0 || !out1 || !dig_idx || !dig_off || (nsp > 0 && !sp_idx))
    return fail(HX_ERR_INVALID, "null argument");
  hx_ctx* c = t0->ctx;
  CTX_ENTER(c);
  if (ndig > W->ndig || ndig < 1)
    return fail(HX_ERR_INVALID, "W must have as many columns as there are digits");
  if ((t1 && (t1->prime_idx != t0->prime_idx || t1->batch != t0->batch)) ||
      t2->prime_idx != t0->prime_idx || t2->batch != t0->batch || out0->batch != t0->batch ||
      out1->batch != t0->batch)
    return fail(HX_ERR_PRIMESET, "reLinearize: parts must share one prime set and batch");
  CHK(check_rows(c, sp_idx, nsp));
  std::vector<int> all;
  CHK(build_all(t0, sp_idx, nsp, all));
  const int L = t0->nrows(), nall = (int)all.size();
  if (nall > MAX_ROWS)
    return fail(HX_ERR_UNSUPPORTED, "too many rows");
  for (int r : all)
    if (find_row(W->row_idx, r) < 0)
      return fail(HX_ERR_PRIMESET, "No key-switching matrix row for prime %d", r);
  size_t rw = t0->row_words();
  // (an output that aliases an input keeps its rows when it has to grow)
  OWN(out0);
  OWN(out1);
  CHK(poly_reserve(out0, nall, out0 == t0 || out0 == t1 || out0 == t2));
  CHK(poly_reserve(out1, nall, out1 == t0 || out1 == t1 || out1 == t2));
  // parts (1),(s): addPrimesAndScale(special) happens inside the key-switch kernel, which reads
  // them unscaled (no separate pass over 2L rows)
  const std::vector<int> own = t0->prime_idx;  // (out0 may be t0 itself)
  out0->prime_idx = all;
  out1->prime_idx = all;
  return relin_core(c, t2->d, own, all, W, dig_idx, dig_off, ndig, t0->batch, out0->d, out1->d, t0->d,
                    t1 ? t1->d : nullptr);
}

// Ctxt::reLinearize with the digit norms keySwitchPart feeds into the noise estimate
// (src/Ctxt.cpp:828-829): norms as in hx_break_into_digits_norms.
extern "C" int hx_relinearize_norms(const hx_poly* t0, const hx_poly* t1, const hx_poly* t2,
                                    const hx_ksk* W, const int* dig_idx, const int* dig_off, int ndig,
                                    const int* sp_idx, int nsp, hx_poly* out0, hx_poly* out1,
                                    double* norms)
{
  if (!t0 || !norms || ndig < 1)
    return fail(HX_ERR_INVALID, "bad argument");
  hx_ctx* c = t0->ctx;
  CTX_ENTER(c);
  NO_CAPTURE(c, "hx_relinearize_norms (a norm read-back)");
  const size_t rw = t0->row_words();
  CHK(frac_begin(c, (size_t)ndig * rw));
  int rc = hx_relinearize(t0, t1, t2, W, dig_idx, dig_off, ndig, sp_idx, nsp, out0, out1);
  return finish_norms(c, rc, (size_t)ndig * rw, ndig * t0->batch, norms, nullptr);
}

// ------------------------------------------------------------------
// HEXL-shim compatibility layer (src/intelExt.h:20-59): host pointers,
// synchronous, one context per (n, q) cached under a mutex like the HEXL NTT
// cache (src/intelExt.cpp:46-73).
//
// Order and root are HEXL's, because the reference's call sites rely on them:
// intel::FFTFwd = hexl::NTT(n, q).ComputeForward (src/intelExt.cpp:76-84)
// delivers out[i] = f(psi^(2*brev(i)+1)) -- BIT-REVERSED evaluation order --
// and Cmodulus::FFT_aux applies BitReverseCopy to it afterwards
// (src/CModulus.cpp:385, :421-426); Cmodulus::iFFT bit-reverses the row before
// intel::FFTRev1 (:510-514), which therefore consumes the same order.  psi is
// the root the NTT object picks for itself: MinimalPrimitiveRoot(2n, q).
// The engine's rows are in natural order (j <-> psi^(2j+1)), so the shim
// permutes on the host -- the call is a PCIe round trip anyway.
// ------------------------------------------------------------------
namespace {
struct ShimEntry {
  hx_ctx* ctx;
  hx_poly* a;
  hx_poly* b;
};
std::mutex g_shim_mu;
std::map<std::pair<long, long>, ShimEntry> g_shim;

int shim_get(long n, long q, ShimEntry** out)
{
  auto key = std::make_pair(n, q);
  auto it = g_shim.find(key);
  if (it == g_shim.end()) {
    if (n < 2 || (n & (n - 1)))
      return fail(HX_ERR_INVALID, "intel:: shim needs a power-of-two n");
    ShimEntry e;
    CHK(hx_ctx_create(&e.ctx, 0, (uint64_t)(2 * n)));
    int idx;
    // (0 when 2n does not divide q-1: hx_ctx_add_prime then reports it)
    const uint64_t root = hxh::hexl_minimal_primitive_root((uint64_t)q, (uint64_t)(2 * n));
    CHK(hx_ctx_add_prime(e.ctx, (uint64_t)q, root, &idx));
    CHK(hx_poly_create(e.ctx, 1, &idx, 1, &e.a));
    CHK(hx_poly_create(e.ctx, 1, &idx, 1, &e.b));
    it = g_shim.emplace(key, e).first;
  }
  *out = &it->second;
  return HX_OK;
}
template <class F>
int shim_unary(long* out, const long* in, long n, long q, F op)
{
  std::lock_guard<std::mutex> lk(g_shim_mu);
  ShimEntry* e;
  CHK(shim_get(n, q, &e));
  CHK(hx_poly_upload(e->a, (const uint64_t*)in));
  CHK(op(e));
  return hx_poly_download(e->a, (uint64_t*)out);
}
template <class F>
int shim_binary(long* r, const long* a, const long* b, long n, long q, F op)
{
  std::lock_guard<std::mutex> lk(g_shim_mu);
  ShimEntry* e;
  CHK(shim_get(n, q, &e));
  CHK(hx_poly_upload(e->a, (const uint64_t*)a));
  CHK(hx_poly_upload(e->b, (const uint64_t*)b));
  CHK(op(e));
  return hx_poly_download(e->a, (uint64_t*)r);
}
}  // namespace

// dst[brev(i)] = src[i] over log2(n) bits (an involution: the same call maps either order to the other)
static void shim_bit_reverse_copy(uint64_t* dst, const uint64_t* src, long n)
{
  int bits = 0;
  while ((1L << bits) < n)
    bits++;
  for (long i = 0; i < n; i++)
    dst[hx::brev_bits((unsigned)i, bits)] = src[i];
}
extern "C" int hx_intel_FFTFwd(long* out, const long* in, long n, long q)
{
  if (!out || !in)
    return fail(HX_ERR_INVALID, "null pointer");
  std::lock_guard<std::mutex> lk(g_shim_mu);
  ShimEntry* e;
  CHK(shim_get(n, q, &e));
  CHK(hx_poly_upload(e->a, (const uint64_t*)in));
  CHK(hx_ntt_forward(e->a));
  std::vector<uint64_t> nat((size_t)n);
  CHK(hx_poly_download(e->a, nat.data()));
  shim_bit_reverse_copy((uint64_t*)out, nat.data(), n);  // natural -> HEXL's bit-reversed output order
  return HX_OK;
}
extern "C" int hx_intel_FFTRev1(long* out, const long* in, long n, long q)
{
  if (!out || !in)
    return fail(HX_ERR_INVALID, "null pointer");
  std::lock_guard<std::mutex> lk(g_shim_mu);
  ShimEntry* e;
  CHK(shim_get(n, q, &e));
  std::vector<uint64_t> nat((size_t)n);
  shim_bit_reverse_copy(nat.data(), (const uint64_t*)in, n);  // HEXL's bit-reversed input order -> natural
  CHK(hx_poly_upload(e->a, nat.data()));
  CHK(hx_ntt_inverse(e->a));
  return hx_poly_download(e->a, (uint64_t*)out);
}
extern "C" int hx_intel_EltwiseAddMod(long* r, const long* a, const long* b, long n, long q)
{
  return shim_binary(r, a, b, n, q, [](ShimEntry* e) { return hx_add(e->a, e->b); });
}
extern "C" int hx_intel_EltwiseSubMod(long* r, const long* a, const long* b, long n, long q)
{
  return shim_binary(r, a, b, n, q, [](ShimEntry* e) { return hx_sub(e->a, e->b); });
}
extern "C" int hx_intel_EltwiseMultMod(long* r, const long* a, const long* b, long n, long q)
{
  return shim_binary(r, a, b, n, q, [](ShimEntry* e) { return hx_mul(e->a, e->b); });
}
extern "C" int hx_intel_EltwiseAddModScalar(long* r, const long* a, long s, long n, long q)
{
  uint64_t sv = (uint64_t)s;
  return shim_unary(r, a, n, q, [sv](ShimEntry* e) { return hx_add_scalar(e->a, &sv); });
}
extern "C" int hx_intel_EltwiseSubModScalar(long* r, const long* a, long s, long n, long q)
{
  uint64_t sv = (uint64_t)s;
  return shim_unary(r, a, n, q, [sv](ShimEntry* e) { return hx_sub_scalar(e->a, &sv); });
}
extern "C" int hx_intel_EltwiseMultModScalar(long* r, const long* a, long s, long n, long q)
{
  uint64_t sv = (uint64_t)s;
  return shim_unary(r, a, n, q, [sv](ShimEntry* e) { return hx_mul_scalar(e->a, &sv); });
}

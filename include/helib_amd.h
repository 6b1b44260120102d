/*
 * helib_amd.h -- C ABI of the MI355X-native DoubleCRT engine.
 *
 * This is the drop-in boundary for HElib 2.2.0's polynomial-arithmetic hot
 * path.  It replaces, one level above HElib's own accelerator seam (the
 * `intel::` HEXL shim, src/intelExt.h:20-59), the following reference
 * interfaces; each entry point below names the one it stands in for:
 *
 *   Cmodulus  ctor / FFT / iFFT        include/helib/CModulus.h:112-145
 *   DoubleCRT storage + ring ops       include/helib/DoubleCRT.h:212-385
 *   DoubleCRT::breakIntoDigits         src/DoubleCRT.cpp:479-561
 *   DoubleCRT::addPrimes / AndScale    src/DoubleCRT.cpp:565-647
 *   DoubleCRT::scaleDownToSet          src/DoubleCRT.cpp:1464-1516
 *   Ctxt::tensorProduct inner loop     src/Ctxt.cpp:1576-1597
 *   Ctxt::keySwitchDigits              src/Ctxt.cpp:191-230
 *   intel::FFTFwd/FFTRev1/Eltwise*     src/intelExt.h:20-59 (compat layer)
 *
 * Conventions
 *   - plain C: opaque handles, raw pointers, sizes.  No exceptions cross the
 *     ABI: every call returns HX_OK (0) or a negative hx_status; the message
 *     of the last failure on the calling thread is hx_last_error().
 *   - a `hx_poly` is a BATCH of `batch` independent DoubleCRT objects that
 *     share one prime set: device layout [row][batch][phi(m)] of uint64
 *     residues in [0, q_row), row r holding prime `prime_idx[r]`
 *     (HElib: IndexMap<vec_long>, one heap vector per prime,
 *     include/helib/DoubleCRT.h:87-95).  batch = 1 is a single DoubleCRT.
 *   - host buffers passed to upload/download use the same [row][batch][N]
 *     order and are owned by the caller; device memory is owned by the
 *     library unless the poly was created with hx_poly_wrap.
 *   - all work of a context is enqueued on one HIP stream (settable); calls
 *     are asynchronous with respect to the host except upload/download/sync.
 *   - thread-safe: every call takes its context's lock while it updates host-side state and
 *     enqueues its kernels (device work is ordered by the context's stream), so several threads
 *     may operate on DISTINCT polys of one context concurrently -- the re-entrancy HElib's NTL
 *     thread pool relies on (src/CModulus.cpp:580-610).  Two threads must not use the SAME poly.
 *   - results are bit-identical to reference HElib's DoubleCRT rows for the
 *     same (q, root) -- values are canonical residues, there is no rounding.
 *   - the library never falls back to the CPU: without a usable gfx950 device
 *     every compute entry point fails with HX_ERR_DEVICE.
 */
#ifndef HELIB_AMD_H
#define HELIB_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hx_ctx hx_ctx;   /* Context (prime chain) + PAlgebra bits used by the path */
typedef struct hx_poly hx_poly; /* batched DoubleCRT                                         */
typedef struct hx_ksk hx_ksk;   /* one KeySwitch matrix resident on the device               */

typedef enum {
  HX_OK = 0,
  HX_ERR_INVALID = -1,     /* bad argument (helib::InvalidArgument)                       */
  HX_ERR_DEVICE = -2,      /* HIP error / no gfx950 device                                  */
  HX_ERR_PRIMESET = -3,    /* index-set mismatch (helib::RuntimeError, DoubleCRT.cpp:243-253)*/
  HX_ERR_NOT_IN_ZMSTAR = -4, /* automorph: k not in Zm* (DoubleCRT.cpp:1166-1167)          */
  HX_ERR_UNSUPPORTED = -5, /* shape not supported by this build                             */
  HX_ERR_NOMEM = -6
} hx_status;

const char* hx_last_error(void);
const char* hx_version(void);
int hx_device_count(int* count);

/* ---------------- context: Context::moduli + zMStar ---------------- */
/* m > 1.  Builds the Z_m^* tables (PAlgebra, src/PAlgebra.cpp:532-538). */
int hx_ctx_create(hx_ctx** out, int device, uint64_t m);
int hx_ctx_destroy(hx_ctx* ctx);
int hx_ctx_phim(const hx_ctx* ctx, uint64_t* phim);
/* stream: a hipStream_t (void*); NULL = the device's default stream. */
int hx_ctx_set_stream(hx_ctx* ctx, void* hip_stream);
int hx_ctx_sync(hx_ctx* ctx);
/* Cmodulus::Cmodulus(zms, q, rt) (src/CModulus.cpp:64-181).
 * root: m = 2^k  -> w0 = NTL RootTable[0][k], a primitive m-th root of unity
 *                   mod q; it is an INPUT because it only exists inside NTL's
 *                   seeded PRG (src/CModulus.cpp:93-119).  root = 0 selects
 *                   FindPrimRootT(q, m) (src/NumbTh.cpp:436-493).
 *       general m -> the FindPrimitiveRoot output of order 2m (m even) / m
 *                   (m odd); 0 = compute it exactly as the reference does.
 * idx_out: position in Context::moduli order (0,1,2,...).                 */
int hx_ctx_add_prime(hx_ctx* ctx, uint64_t q, uint64_t root, int* idx_out);
int hx_ctx_num_primes(const hx_ctx* ctx, int* n);
int hx_ctx_prime(const hx_ctx* ctx, int idx, uint64_t* q, uint64_t* root);

/* ---------------- DoubleCRT storage ---------------- */
/* DoubleCRT(context, indexSet): zero-initialised rows for primes prime_idx[]. */
int hx_poly_create(hx_ctx* ctx, int batch, const int* prime_idx, int nrows, hx_poly** out);
/* Same without the zero fill, for objects that are about to be overwritten (outputs). */
int hx_poly_create_uninit(hx_ctx* ctx, int batch, const int* prime_idx, int nrows, hx_poly** out);
/* Same, on caller-owned device memory (e.g. a torch tensor's data_ptr) of
 * nrows*batch*phim uint64; not zeroed, never freed by the library. */
int hx_poly_wrap(hx_ctx* ctx, int batch, const int* prime_idx, int nrows, void* device_ptr,
                 hx_poly** out);
int hx_poly_destroy(hx_poly* p);
int hx_poly_shape(const hx_poly* p, int* batch, int* nrows, uint64_t* phim);
int hx_poly_primes(const hx_poly* p, int* prime_idx_out /* nrows */);
void* hx_poly_device_ptr(hx_poly* p);
int hx_poly_upload(hx_poly* p, const uint64_t* host);         /* synchronous */
int hx_poly_download(const hx_poly* p, uint64_t* host);       /* synchronous */
/* DoubleCRT::operator= (src/DoubleCRT.cpp:815-837): dst takes src's prime set
 * (dst capacity must be >= src rows). */
int hx_poly_copy(hx_poly* dst, const hx_poly* src);
int hx_poly_set_zero(hx_poly* p);
/* DoubleCRT::randomize (src/DoubleCRT.cpp:1258-1378): every row of every batch element filled with
 * uniform residues by the reference's rejection sampling (2048-byte buffers, ceil(k/8) bytes per
 * candidate, little endian, masked to k = NumBits(q-1) bits, kept when < q) -- on the device, from
 * a ChaCha20 (RFC 8439) key stream per row: 256-bit key `key32`, nonce = (stream low word, stream
 * high word, prime index | batch element << 16), block counter from 0.  NTL's RandomStream (the
 * reference's source of bytes) cannot be reproduced without NTL; the sampling rule is the
 * reference's, the stream is this library's (known-answer tests: RFC 8439 2.3.2 + the oracle). */
int hx_randomize(hx_poly* p, const uint8_t* key32, uint64_t stream);
/* DoubleCRT::removePrimes: metadata only (rows are compacted on the device). */
int hx_poly_remove_primes(hx_poly* p, const int* prime_idx, int n);

/* ---------------- Cmodulus::FFT / iFFT over all rows ---------------- */
/* In place: coefficient rows (deg < phim, reduced mod q) <-> evaluation rows
 * at the primitive m-th roots, natural (Z_m^*) order.
 * src/CModulus.cpp:358-444 / :486-578 ; DoubleCRT::FFT src/DoubleCRT.cpp:68-105 */
int hx_ntt_forward(hx_poly* p);
int hx_ntt_inverse(hx_poly* p);

/* ---------------- DoubleCRT element-wise ring ops ---------------- */
/* a op= b on the rows of a; requires primes(a) subset of primes(b), otherwise
 * HX_ERR_PRIMESET (DoubleCRT::Op, src/DoubleCRT.cpp:216-273, do_mul :278-337).
 * batch(b) must equal batch(a) or be 1 (broadcast). */
int hx_add(hx_poly* a, const hx_poly* b);
int hx_sub(hx_poly* a, const hx_poly* b);
int hx_mul(hx_poly* a, const hx_poly* b);
int hx_negate(hx_poly* a); /* DoubleCRT::Negate src/DoubleCRT.cpp:363-384 */
/* a op= scalar, scalar given per row already reduced mod that row's prime
 * (DoubleCRT::Op(ZZ) src/DoubleCRT.cpp:339-361). */
int hx_add_scalar(hx_poly* a, const uint64_t* c_per_row);
int hx_sub_scalar(hx_poly* a, const uint64_t* c_per_row);
int hx_mul_scalar(hx_poly* a, const uint64_t* c_per_row);
/* DoubleCRT::operator=(ZZ) (src/DoubleCRT.cpp:866-884): every entry of row r = c_per_row[r] mod q_r */
int hx_set_scalar(hx_poly* a, const uint64_t* c_per_row);
/* DoubleCRT::Exp (src/DoubleCRT.cpp:1142-1156): entry-wise PowerMod(x, e, q_r), e >= 0 */
int hx_exp(hx_poly* a, uint64_t e);
/* DoubleCRT::automorph(k) (src/DoubleCRT.cpp:1160-1202); in place. */
int hx_automorph(hx_poly* a, uint64_t k);
/* DoubleCRT::complexConj (src/DoubleCRT.cpp:1240-1255) */
int hx_complex_conj(hx_poly* a);

/* ---------------- exact RNS basis operations ---------------- */
/* DoubleCRT::addPrimesAndScale (src/DoubleCRT.cpp:603-647): multiply rows by
 * prod(add_idx) and append zero rows for add_idx (capacity permitting). */
int hx_add_primes_and_scale(hx_poly* a, const int* add_idx, int nadd);
/* DoubleCRT::addPrimes (src/DoubleCRT.cpp:565-599): exact centred extension of
 * the RNS basis by add_idx (toPoly + FFT on the new primes), in place. */
int hx_add_primes(hx_poly* a, const int* add_idx, int nadd);
/* DoubleCRT::toPoly (src/DoubleCRT.cpp:925-1113: iFFT of every row, CRT, centring) followed by
 * PolyRed(poly, t, abs=true) (src/NumbTh.cpp:775-803), the tail of SecKey::Decrypt
 * (src/keys.cpp:1383-1405): out_host[b*phi(m) + j] = centred coefficient j of batch element b,
 * reduced into [0,t).  Exact (mixed-radix digits on the device, no big integers); `a` is unchanged.
 * t in [2, 2^60); at most 64 rows.  Synchronous (the result is on the host when it returns). */
int hx_poly_rem(const hx_poly* a, uint64_t t, uint64_t* out_host);
/* DoubleCRT::scaleDownToSet (src/DoubleCRT.cpp:1464-1516): drop drop_idx with
 * exact rounding, delta forced to 0 mod ptxt_space.  In place as far as the caller can tell: the
 * kept rows keep their order; a library-owned poly may move to another slab (hx_poly_device_ptr
 * changes), a poly made with hx_poly_wrap always has its result in the caller's buffer. */
int hx_scale_down(hx_poly* a, const int* drop_idx, int ndrop, uint64_t ptxt_space);
/* The same for several DoubleCRT objects sharing one prime set and batch (all parts of the
 * ciphertexts of one Ctxt::modDownToSet, src/Ctxt.cpp:462-465) in one pair of launches. */
int hx_scale_down_multi(hx_poly** polys, int npoly, const int* drop_idx, int ndrop,
                        uint64_t ptxt_space);
/* Ctxt::bringToSet (src/Ctxt.cpp:373-389) on several parts: modUpToSet by add_idx
 * (addPrimesAndScale) followed by modDownToSet dropping drop_idx; one fused pair of launches
 * when a single prime is dropped. */
int hx_bring_to_set_multi(hx_poly** polys, int npoly, const int* add_idx, int nadd,
                          const int* drop_idx, int ndrop, uint64_t ptxt_space);
/* Ctxt::tensorProduct of two canonical two-part ciphertexts (src/Ctxt.cpp:1563-1608) immediately followed by
 * Ctxt::bringToSet of the product -- what Ctxt::multiplyBy does between multLowLvl's tensor product and the key
 * switch (reLinearize -> dropSmallAndSpecialPrimes, src/Ctxt.cpp:720-760).  c0, c1 / d0, d1: the parts (1), (s) of
 * the two operands on one prime set; o0, o1, o2 receive the product parts (1), (s), (s^2) on that set with add_idx
 * added and drop_idx dropped (their previous contents and prime sets are irrelevant; they must not alias an operand).
 * On a power-of-two ring (N = 2^13..2^15) the product is formed inside the mod-down kernels from the operands' rows
 * and never written on the old prime set -- one dropped prime (a fresh multiply) and several (every later multiply)
 * alike; any other shape runs hx_tensor and hx_bring_to_set_multi one after the other.  Results are identical to
 * that sequence word for word.  _norms: as hx_bring_to_set_multi_norms, three
 * polys. */
int hx_tensor_bring_to_set(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1,
                           hx_poly* o0, hx_poly* o1, hx_poly* o2, const int* add_idx, int nadd,
                           const int* drop_idx, int ndrop, uint64_t ptxtSpace);
int hx_tensor_bring_to_set_norms(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1,
                                 hx_poly* o0, hx_poly* o1, hx_poly* o2, const int* add_idx, int nadd,
                                 const int* drop_idx, int ndrop, uint64_t ptxtSpace, double* norms);

/* DoubleCRT::breakIntoDigits (src/DoubleCRT.cpp:479-561).  a has ctxt primes
 * only; digit d = dig_idx[dig_off[d]..dig_off[d+1]); special primes sp_idx.
 * digits_out: poly with ndig*(nrows(a)+nsp) rows, block d holding digit d on
 * primes(a) followed by sp_idx. */
int hx_break_into_digits(const hx_poly* a, const int* dig_idx, const int* dig_off, int ndig,
                         const int* sp_idx, int nsp, hx_poly* digits_out);

/* ---------------- environment switches ----------------
 * None is needed in production: each selects the older or the generic form of a path, for tests that must reach it
 * and for same-box A/B measurements; results are bit-identical either way.  The library snapshots them when a
 * context is created (hx_ctx_create; helib_amd/csrc/switches.h is the one place that reads the environment).
 *   HX_NO_HPS, HX_HPS_EPS=x, HX_HPS_MIN_N=n   exact-RNS kernels: Garner instead of the HPS front end; the trust
 *                                            margin of an HPS quotient (default 2^-30); HPS from n sources on (9)
 *   HX_NO_LAZY_RNS                            no 128-bit lazy sums / single-subtraction Garner steps
 *   HX_NO_FAST_BREAK, HX_NO_FAST_EXTEND, HX_NO_WIDE_EXTEND   generic breakIntoDigits / basis-extension kernels
 *   HX_NO_TENSOR_MULTI, HX_NO_MULRELIN_FUSE   tensor product as a pass of its own in front of the several-primes
 *                                            mod-switch / inside hx_mul_relin
 *   HX_NO_PROTH                               row transforms: Shoup butterflies on every row (by default rows of primes
 *                                            q = 1 mod 2^32 -- every chain prime of the benchmarks -- run the Proth-form ones)
 *   HX_NO_PROTH_RNS                           exact-RNS kernels: Barrett / Shoup products on Proth-form primes too (by default
 *                                            their Garner steps, target sums and fix-ups are Montgomery products; HX_NO_PROTH
 *                                            implies it)
 *   HX_BLUE_OLD                               general m: the chain of passes instead of one convolution kernel
 *   HX_NO_PFA                                 m = 21845: Bluestein instead of the Good-Thomas x Rader kernels
 *   HX_PFA_NO_REM                             ... their inverse without the fused rem Phi_m (two convolution launches instead)
 *   HX_NORM_ASYNC, HX_NORM_OLD, HX_NORM_PLAIN, HX_NORM_MEMCPY
 *                                            variants of the canonical-embedding norm kernels and their read-back
 *   HX_WAIT_POLL_US=n                         how long a norm read-back is polled for before the thread sleeps (2000)
 *   HX_ARENA_TRACE                            one line on stderr per hipMalloc of the slab arena
 * (include/helib_amd_ctxt.hpp, the C++ host: HX_NO_LAZY_TENSOR -- multiplyBy forms the tensor product eagerly.) */

/* ---------------- ciphertext-level fused loops ---------------- */
/* KeySwitch matrix W (include/helib/keySwitching.h:86-101) with the a-column
 * expanded once by the host (HElib regenerates it from W.prgSeed on every key
 * switch, src/Ctxt.cpp:196-206).  b, a: host arrays [ndig][nrows][phim] on
 * primes row_idx[nrows] (ctxt primes followed by special primes). */
int hx_ksk_create(hx_ctx* ctx, int ndig, const int* row_idx, int nrows, const uint64_t* b,
                  const uint64_t* a, hx_ksk** out);
int hx_ksk_destroy(hx_ksk* k);
/* the matrix back on the host, for a checker or for another process that is to hold the same key (one key pair
 * replicated over the GPUs of a node): shape first (row_idx_out may be NULL), then b, a =
 * [ndig][nrows][phim] as hx_ksk_create took them (KeySwitch::b / the expanded a column, include/helib/keySwitching.h:86-101) */
int hx_ksk_shape(const hx_ksk* k, int* ndig, int* nrows, int* row_idx_out /* nrows */);
int hx_ksk_download(const hx_ksk* k, uint64_t* b, uint64_t* a);   /* synchronous */

/* Ctxt::tensorProduct for two 2-part ciphertexts (src/Ctxt.cpp:1576-1597):
 * o0 = c0*d0, o1 = c0*d1 + c1*d0, o2 = c1*d1. */
int hx_tensor(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1,
              hx_poly* o0, hx_poly* o1, hx_poly* o2);
/* Ctxt::keySwitchDigits (src/Ctxt.cpp:191-230):
 * out0 += sum_d digit_d*b_d ; out1 += sum_d digit_d*a_d */
int hx_key_switch_digits(const hx_poly* digits, const hx_ksk* W, hx_poly* out0, hx_poly* out1);

/* Ctxt::multiplyBy data path at a fixed level: tensorProduct + reLinearize
 * (src/Ctxt.cpp:1563-1608, :720-842).  Inputs: 2-part ciphertexts (c0,c1),
 * (d0,d1) on the same ctxt primes; outputs (out0,out1) on ctxt ∪ special
 * primes (W's row set), exactly what reLinearize leaves in the Ctxt.
 * Digits as in hx_break_into_digits. */
int hx_mul_relin(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1,
                 const hx_ksk* W, const int* dig_idx, const int* dig_off, int ndig,
                 hx_poly* out0, hx_poly* out1);
/* ... with the digit norms of hx_relinearize_norms: norms[d * batch + b] (what keySwitchPart multiplies by the
 * matrix' noise bound, src/Ctxt.cpp:828-841) */
int hx_mul_relin_norms(const hx_poly* c0, const hx_poly* c1, const hx_poly* d0, const hx_poly* d1,
                       const hx_ksk* W, const int* dig_idx, const int* dig_off, int ndig,
                       hx_poly* out0, hx_poly* out1, double* norms);

/* Ctxt::reLinearize of a 3-part ciphertext (1, s, s^2) (src/Ctxt.cpp:720-786, keySwitchPart
 * :805-842): parts (1),(s) get addPrimesAndScale(special), part s^2 is broken into digits and
 * multiplied by W.  W may cover more ctxt primes than the parts currently have (lower level);
 * digits = the context's digits restricted to the parts' primes (leading digits of W).
 * out0/out1: primes(t0) followed by sp_idx.
 * t1 may be NULL: no part points at s -- the (1, s(X^k)) ciphertext that Ctxt::smartAutomorph
 * relinearises after Ctxt::automorph (src/Ctxt.cpp:2437-2515), t2 then being the s(X^k) part and
 * W the matrix for that automorphism. */
int hx_relinearize(const hx_poly* t0, const hx_poly* t1, const hx_poly* t2, const hx_ksk* W,
                   const int* dig_idx, const int* dig_off, int ndig, const int* sp_idx, int nsp,
                   hx_poly* out0, hx_poly* out1);

/* ---------------- measured noise: canonical-embedding norms ----------------
 * embeddingLargestCoeff (src/norms.cpp:129-262,480-493): max over j in Z_m^* of |f(W^j)|,
 * W = exp(2 pi i/m), evaluated on the device in double precision (the reference uses PGFFT,
 * src/PGFFT.cpp); parity is to a relative tolerance of 1e-9.  m a power of two, or any
 * m <= 131072 (complex-double Bluestein); beyond that HX_ERR_UNSUPPORTED: the host keeps the
 * reference's high-probability bound, src/DoubleCRT.cpp:520-529.  These calls synchronise the stream: the numbers land in host
 * memory.  The arithmetic results are exactly those of the plain calls. */
/* Deferred read-back: after hx_ctx_defer_norms(ctx, 1) the *_norms calls do not synchronise;
 * their `norms` arrays (which must stay valid) are filled by hx_norms_flush(ctx), which waits only
 * for the norm kernels already enqueued, not for work enqueued after them -- the host can keep
 * the GPU fed while it waits for the numbers its next prime-set decision needs.
 * hx_ctx_defer_norms(ctx, 0) flushes and returns to synchronous behaviour. */
int hx_ctx_defer_norms(hx_ctx* ctx, int on);
int hx_norms_flush(hx_ctx* ctx);
/* rows real polynomials of phi(m) coefficients each (host) -> norms_out[rows] (host) */
int hx_embedding_norm(hx_ctx* ctx, const double* f_host, int rows, double* norms_out);
/* hx_scale_down_multi + norms[npoly*batch] = embeddingLargestCoeff(fdelta) with
 * fdelta = delta/diffProd (src/Ctxt.cpp:466-507); fdelta (optional, host,
 * [npoly][batch][phim]) receives the coefficients themselves. */
int hx_scale_down_multi_norms(hx_poly** polys, int npoly, const int* drop_idx, int ndrop,
                              uint64_t ptxt_space, double* norms, double* fdelta);
int hx_bring_to_set_multi_norms(hx_poly** polys, int npoly, const int* add_idx, int nadd,
                                const int* drop_idx, int ndrop, uint64_t ptxt_space,
                                double* norms);
/* hx_break_into_digits + its return value (src/DoubleCRT.cpp:538-545) in pieces:
 * norms[d*batch+b] = embeddingLargestCoeff(digit d of element b) / P_d, P_d = product of the
 * digit's primes (multiplied back by the host in extended range). */
int hx_break_into_digits_norms(const hx_poly* a, const int* dig_idx, const int* dig_off, int ndig,
                               const int* sp_idx, int nsp, hx_poly* digits_out, double* norms);
/* hx_relinearize + the digit norms Ctxt::keySwitchPart multiplies by W.noiseBound
 * (src/Ctxt.cpp:828-829); layout as above. */
int hx_relinearize_norms(const hx_poly* t0, const hx_poly* t1, const hx_poly* t2, const hx_ksk* W,
                         const int* dig_idx, const int* dig_off, int ndig, const int* sp_idx,
                         int nsp, hx_poly* out0, hx_poly* out1, double* norms);

/* ---------------- HEXL-shim compatibility layer ---------------- */
/* Same signatures and semantics as namespace intel (src/intelExt.h:20-59):
 * host pointers, synchronous, in-place allowed.  FFTFwd / FFTRev1 are what
 * hexl::NTT(n, q).ComputeForward / ComputeInverse are (src/intelExt.cpp:76-98):
 *   - the root is the NTT object's own, MinimalPrimitiveRoot(2n, q), the smallest
 *     primitive 2n-th root of unity (no root crosses this seam);
 *   - FFTFwd returns BIT-REVERSED evaluation order, out[i] = f(psi^(2*brev(i)+1)),
 *     and FFTRev1 reads that order.  The reference's call sites depend on it:
 *     Cmodulus::FFT_aux runs BitReverseCopy after intel::FFTFwd
 *     (src/CModulus.cpp:385, :421-426) and Cmodulus::iFFT before intel::FFTRev1
 *     (:510-514), which makes the stored row the natural one, y[j] = f(psi^(2j+1)),
 *     that DoubleCRT::automorph (src/DoubleCRT.cpp:1160-1202) and the wire format
 *     index.  (hx_ntt_forward / hx_ntt_inverse work on natural rows directly.)
 * One PCIe round trip per call -- provided for link compatibility of a
 * USE_INTEL_HEXL-style build, not for speed. */
int hx_intel_FFTFwd(long* out, const long* in, long n, long q);
int hx_intel_FFTRev1(long* out, const long* in, long n, long q);
int hx_intel_EltwiseAddMod(long* r, const long* a, const long* b, long n, long q);
int hx_intel_EltwiseAddModScalar(long* r, const long* a, long scalar, long n, long q);
int hx_intel_EltwiseSubMod(long* r, const long* a, const long* b, long n, long q);
int hx_intel_EltwiseSubModScalar(long* r, const long* a, long scalar, long n, long q);
int hx_intel_EltwiseMultMod(long* r, const long* a, const long* b, long n, long q);
int hx_intel_EltwiseMultModScalar(long* r, const long* a, long scalar, long n, long q);

/* ---------------- measurement helpers ---------------- */
/* Runs `iters` back-to-back launches of the forward (dir=0) or inverse (dir=1)
 * NTT kernel on the first max_rows rows of p (0 = all rows) between two HIP
 * events recorded on the context's stream and returns the average kernel time
 * in milliseconds (bench.py roofline leg). */
int hx_time_ntt(hx_poly* p, int dir, int iters, int max_rows, float* avg_ms);
/* HIP events on the context's stream: begin records one, end records the second, waits for it
 * and returns the elapsed milliseconds of everything enqueued on the context in between. */
int hx_ctx_timer_begin(hx_ctx* ctx);
int hx_ctx_timer_end(hx_ctx* ctx, float* ms);
/* In-situ kernel timing (the reference's counterpart is its FHE timers printed by printAllTimers,
 * src/timing.cpp:87; here the unit is a device kernel).  Between begin and end every kernel the
 * library launches, on any context of the process, is bracketed by a pair of HIP events recorded on
 * the stream it is launched on, i.e. it is timed where it runs inside the real sequence.
 * hx_profile_end waits for the recorded launches and writes a NUL-terminated JSON summary
 *   {"launches": n, "dropped": d, "kernels": [{"kernel": name, "workgroups": g, "workgroup_size": t,
 *     "calls": k, "total_us": .., "avg_us": .., "min_us": .., "max_us": ..}, ... by total time]}
 * into json[0..cap).  With json == NULL it only reports the size in *needed (the summary is kept
 * for the next call).  Launches recorded into a HIP graph are not timed. */
int hx_profile_begin(void);
int hx_profile_end(char* json, size_t cap, size_t* needed);
/* The device-memory arena behind the context's DoubleCRT slabs (HElib leaves this to malloc through
 * NTL's vec_long, include/helib/DoubleCRT.h:87-95): out[0] = bytes reserved from hipMalloc, out[1] =
 * bytes handed out to polys, out[2] = hipMalloc calls made so far (a warm loop adds none, whatever it
 * keeps alive), out[3] = blocks parked because a live HIP graph may still point at them. */
int hx_ctx_arena_stats(hx_ctx* ctx, uint64_t out[4]);
/* Reserves at least `bytes` of device memory for the context's slabs now (one hipMalloc for what is missing), so
 * that a loop whose footprint is known -- a benchmark's batch, a pipeline's working set -- never reaches hipMalloc
 * while it runs: a multi-GiB hipMalloc takes tens of milliseconds and stalls the stream.  Sized for 288 GB parts:
 * reserve generously. */
int hx_ctx_reserve(hx_ctx* ctx, uint64_t bytes);

/* ---- HIP graphs: the launch-bound case -------------------------------------------------------
 * The reference's benchmark loop runs ONE ciphertext at a time (benchmarks/bgv_basic.cpp:158-164:
 * copy, multiplyBy); on the device that is ~40 kernels of a few microseconds each, bound by launch
 * latency.  hx_ctx_graph_begin starts recording everything enqueued on the context (the calls
 * return as usual but nothing runs), hx_ctx_graph_end closes the recording into a graph, and
 * hx_graph_launch replays it with one launch: the same kernels on the same buffers -- inputs are
 * whatever the input polys hold at launch time, results land in the polys the recorded calls
 * produced (keep them).  Run the sequence once eagerly first (plans, tables and kernel attributes
 * are set up on first use).  Calls that must wait for the device -- uploads, downloads, norm
 * read-backs (use the reference's noise bounds, not measured noise, in a captured sequence) --
 * cannot be recorded and make hx_ctx_graph_end fail.  While a graph is alive the context keeps
 * every buffer it may reference. */
typedef struct hx_graph hx_graph;
int hx_ctx_graph_begin(hx_ctx* ctx);
int hx_ctx_graph_end(hx_ctx* ctx, hx_graph** out);
int hx_graph_launch(hx_graph* g);
int hx_graph_destroy(hx_graph* g);

#ifdef __cplusplus
}
#endif
#endif /* HELIB_AMD_H */

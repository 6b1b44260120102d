/* helib_amd_host.h -- C ABI of the C++17 host library (helib_amd/lib/libhelib_amd_host.so).
 *
 * The north star keeps the host in C++17 with HElib's Ctxt / DoubleCRT / KeySwitch surface
 * (include/helib_amd_ctxt.hpp, helib_amd_keys.hpp over include/helib_amd.h).  This library is that
 * host compiled once, behind a handful of C entry points, so that a driver written in any language
 * (bench.py, a cgo / JNI / ctypes stub) runs the reference's benchmark loops with the C++ host on the
 * timed path -- the loops of benchmarks/bgv_basic.cpp:144-165 and benchmarks/ckks_basic.cpp:161-180:
 *
 *     ContextBuilder<BGV>().m(m).p(p).r(r).bits(bits).c(3)          (scheme 0)
 *     ContextBuilder<CKKS>().m(m).precision(r).bits(bits).scale(10).c(3)   (scheme 1; benchmarks/ckks_common.h:45-50)
 *     SecKey::GenSecKey + the relinearisation matrix (addSome1DMatrices is not needed for multiplyBy)
 *     two public-key encryptions of random plaintexts               (x batch, packed along the batch axis)
 *     loop:  copy = ctxt1;  copy.multiplyBy(ctxt2);                 (hxh_multiply, level 1)
 *            copy = prod;   copy.multiplyBy(prod);                  (level 2: operands that carry the
 *                                                                    special primes of a key switch)
 *
 * Every function returns 0 or a negative code with the message in hxh_last_error() (thread-local);
 * no exception crosses the ABI.  The device work is enqueued on `stream` (a hipStream_t; NULL = the
 * default stream): the caller brackets its timed region with its own synchronisation.
 */
#ifndef HELIB_AMD_HOST_H
#define HELIB_AMD_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hxh_session hxh_session;

/* Builds the context, the key pair with its relinearisation matrix, and `batch` pairs of fresh
 * encryptions of seeded random plaintexts (BGV: uniform residues mod p^r; CKKS: real coefficients uniform in
 * +-1 / (8 sqrt(phi(m)/3)) -- canonical embedding below the declared size 1 -- encoded at the factor
 * PubKey::Encrypt(Ptxt<CKKS>) uses, EncryptedArrayCx::encodeScalingFactor(): 2^11 at m = 65536, precision(1)).
 * seed = 0: key material from OS entropy (the plaintexts stay seeded). */
int hxh_session_create(hxh_session** out, int device, void* stream, int scheme, long m, long p, long r,
                       long bits, int batch, uint64_t seed);
int hxh_session_destroy(hxh_session* s);
/* info[0..7] = phi(m), #ctxt primes, #special primes, #digits, #small primes, bits of the first ctxt prime,
 * bits of the first special prime, batch */
int hxh_session_info(const hxh_session* s, long info[8]);

/* k x [copy(a); copy.multiplyBy(b)] enqueued back to back; level 1: a, b = the two fresh ciphertexts,
 * level 2: a = b = the product kept by the last level-1 call.  measure = 1: added noise measured on the
 * device as in the reference's default build, read back lazily (each result's estimate is completed
 * one multiply later); 0: the reference's alternative bounds.  The last product is kept for
 * hxh_decrypt / the next level.  Returns when everything is enqueued and the estimates are read. */
int hxh_multiply(hxh_session* s, int level, int k, int measure);
/* one multiply of batch element 0 alone (the reference's own loop shape: one ciphertext at a time) */
int hxh_multiply_single(hxh_session* s, int measure);

/* the plaintexts: out[b*phi + j], which = 0 / 1.  BGV: residues in [0, p^r); CKKS: the encoded reals
 * rint(v * f) / f, f = encodeScalingFactor() */
int hxh_plaintext(const hxh_session* s, int which, double* out);
/* decrypts batch element b of the kept product of `level` (0 = the first fresh ciphertext itself):
 * BGV: phi(m) residues; CKKS: phi(m) decoded reals (raw / ratFactor).  bound (optional): CKKS error
 * bound the ciphertext reports (noiseBound / ratFactor); BGV: log2 of the remaining capacity */
int hxh_decrypt(hxh_session* s, int level, int b, double* out, double* bound);
/* prime indices of the kept product of `level` (ascending); returns the count through *n (cap = size of out) */
int hxh_result_primes(const hxh_session* s, int level, int* out, int cap, int* n);

/* ---- one key pair, many GPUs: rank 0's key material replicated, every rank encrypts and multiplies
 * its own shard under it.  hxh_export_keys: the key pair with its relinearisation matrix as 64-bit words (size
 * query with out = NULL); hxh_session_create_with_keys: a session of the same parameters whose key pair is that
 * material -- enc_seed drives this session's encryption randomness and plaintexts only. ---- */
int hxh_export_keys(hxh_session* s, uint64_t* out, size_t cap_words, size_t* need_words);
int hxh_session_create_with_keys(hxh_session** out, int device, void* stream, int scheme, long m, long p, long r,
                                 long bits, int batch, uint64_t enc_seed, const uint64_t* keys, size_t key_words);

/* ---- ciphertexts across processes (SURVEY 2.3 row C1 "batch scatter/gather", 8e): the service shape of BASELINE
 * configs[3] -- the ciphertext pairs arrive at one place, the products leave from one place, the GPUs in between hold
 * PUBLIC key material only.  A ciphertext crosses a process boundary in the reference's own binary format
 * (Ctxt::writeTo / Ctxt::read, src/Ctxt.cpp:2584-2641; include/helib_amd_wire.hpp).
 *   hxh_session_create_source   a session that only HOLDS `batch` pairs (keys, plaintexts, encryptions; no arena
 *                               reservation for a multiply loop): rank 0 of a scatter
 *   hxh_export_public_keys      the key material without the secret polynomial (SecKey::exportKeys(false))
 *   hxh_export_ctxts            batch elements [first, first + count) of operand `which` (level 0) or of the kept product
 *                               (level 1 / 2), one Ctxt::writeTo blob after the other (size query with out = NULL)
 *   hxh_session_create_from_ctxts  a worker: public keys + two blobs of `batch` ciphertexts as its operands; it
 *                               multiplies, it cannot decrypt (hxh_decrypt / hxh_plaintext fail on it)
 *   hxh_decrypt_wire            ONE ciphertext of a blob under this session's secret key (BGV residues / CKKS decoded
 *                               reals as hxh_decrypt); *used = bytes consumed ---- */
int hxh_session_create_source(hxh_session** out, int device, void* stream, int scheme, long m, long p, long r, long bits,
                              int batch, uint64_t seed);
int hxh_export_public_keys(hxh_session* s, uint64_t* out, size_t cap_words, size_t* need_words);
int hxh_export_ctxts(hxh_session* s, int level, int which, int first, int count, uint8_t* out, size_t cap_bytes,
                     size_t* need_bytes);
int hxh_session_create_from_ctxts(hxh_session** out, int device, void* stream, int scheme, long m, long p, long r, long bits,
                                  int batch, const uint64_t* keys, size_t key_words, const uint8_t* a, size_t a_bytes,
                                  const uint8_t* b, size_t b_bytes);
int hxh_decrypt_wire(hxh_session* s, const uint8_t* blob, size_t bytes, double* out, double* bound, size_t* used);

/* ---- what a checker needs of a session: the chain, the ciphertexts' rows and bookkeeping, the matrix ---- */
/* the chain's primes in Context::moduli order (small, ctxt, special) */
int hxh_chain_primes(const hxh_session* s, uint64_t* out, int cap, int* n);
/* a ciphertext the session holds: level 0 = fresh operand `which` (0 / 1), level 1 / 2 = the kept product;
 * info = lnNoise, lnRatFactor, ptxtMag, intFactor, ptxtSpace, #parts, the key's ln noiseBound and ptxtSpace */
int hxh_ctxt_info(hxh_session* s, int level, int which, double info[8]);
/* part 0 / 1 (handles 1, s) of it: out = [row][batch][phi(m)] words (NULL: shape only), idx_out = the rows' primes */
int hxh_ctxt_rows(hxh_session* s, int level, int which, int part, uint64_t* out, int* idx_out, int cap_rows, int* nrows);
/* the relinearisation matrix: b, a = [ndig][nrows][phi(m)] (NULL: shape only), idx_out = the rows' primes */
int hxh_relin_matrix(hxh_session* s, uint64_t* b, uint64_t* a, int* idx_out, int cap_rows, int* ndig, int* nrows);
/* hx_ctx_arena_stats of the session's device context: reserved bytes, bytes in use, hipMalloc calls, chunks */
int hxh_arena_stats(hxh_session* s, uint64_t out[4]);
/* SecKey::EncryptBatch / DecryptBatch (include/helib_amd_keys.hpp) over `batch` random plaintexts, `reps` times each,
 * timed in the C++ host with the device drained: out = {ms per ciphertext encrypted, ms per ciphertext decrypted, batch,
 * 1.0 when every element decrypted to its plaintext}.  BGV sessions.  (benchmarks/bgv_basic.cpp:186-211 time one
 * ciphertext per call; this is the batched counterpart the engine's batch axis allows.) */
int hxh_encrypt_decrypt_batch(hxh_session* s, int batch, int reps, double out[4]);

const char* hxh_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* HELIB_AMD_HOST_H */

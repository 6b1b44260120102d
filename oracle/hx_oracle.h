/*
 * hx_oracle.h -- CPU restatement of the HElib 2.2.0 DoubleCRT hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke()
 * entry of __graft_entry__.py and the cpu_baseline leg of bench.py may load
 * it.  The shipped path (helib_amd/, include/helib_amd.h) never links,
 * imports or calls anything in this directory.
 *
 * Every function cites the reference file:line (relative to the HElib 2.2.0
 * tree) whose algorithm it restates.  The reference cannot be compiled in
 * this environment (NTL 11.5.1 / GMP are un-vendored, un-installed
 * dependencies), so the arithmetic NTL supplies (FFTFwd/FFTRev1, fftRep
 * convolution, MulMod, ZZ CRT) is restated from its published definition:
 * every value here is a canonical residue in [0,q), which is mathematically
 * determined once (q, root, evaluation order) are fixed.
 *
 * Parity status:
 *   - general m (Bluestein convention, root from FindPrimRootT): PINNED by
 *     the reference-authored fixture tests/test_resources/iotest_asciiLE.txt
 *     (see tests/golden/ and tests/test_oracle_golden.py).
 *   - m = 2^k: the root w0 is NTL's RootTable[0][k], which exists only inside
 *     NTL's seeded PRG (src/CModulus.cpp:93-119).  It is an INPUT here; when
 *     the caller passes root=0 the oracle substitutes FindPrimRootT(q, m).
 *     Value-level parity with an NTL build for that root is "parity unpinned".
 */
#ifndef HX_ORACLE_H
#define HX_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- modular primitives (NTL::MulMod/PowerMod/InvMod semantics) ---- */
uint64_t ho_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t ho_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t ho_invmod(uint64_t a, uint64_t q); /* any modulus, gcd(a,q)=1 */
int ho_is_prime(uint64_t n);                /* deterministic for n < 2^64 */

/* ---- PrimeGenerator  (src/PrimeGenerator.h:41-126) ---- */
typedef struct {
  long len, m, k, t;
} ho_primegen;
void ho_primegen_init(ho_primegen* g, long len, long m);
long ho_primegen_next(ho_primegen* g); /* 0 when it runs out of primes */

/* ---- FindPrimRootT  (src/NumbTh.cpp:436-493) ---- */
uint64_t ho_find_prim_root(uint64_t q, uint64_t e);

/* ---- Intel HEXL's negacyclic NTT, as the intel:: seam calls it (src/intelExt.cpp:76-98) ----
 * HEXL is a third-party dependency absent from /root/reference (HElib asks for >= 1.2.1,
 * CMakeLists.txt:214); this restates its PUBLISHED reference algorithm (hexl/ntt/ntt-internal:
 * ReferenceForwardTransformToBitReverse / ReferenceInverseTransformFromBitReverse, the radix-2
 * Cooley-Tukey / Gentleman-Sande networks over RootOfUnityPowers in bit-reversed order) and its root
 * rule (hexl/number-theory: MinimalPrimitiveRoot(2n, q) = the smallest primitive 2n-th root).  Parity is
 * anchored on the reference's own call sites: forward output is consumed by BitReverseCopy
 * (src/CModulus.cpp:385, :421-426), inverse input is produced by it (:510-514).
 *   forward:  out[i] = sum_k in[k] * psi^(k * (2*brev(i)+1))      (bit-reversed evaluation order)
 *   inverse:  its inverse, reading that order, returning natural coefficients (1/n included) */
uint64_t ho_hexl_minimal_primitive_root(uint64_t q, uint64_t e);
void ho_hexl_forward(uint64_t* out, const uint64_t* in, long n, uint64_t q);
void ho_hexl_inverse(uint64_t* out, const uint64_t* in, long n, uint64_t q);

/* ---- Z_m^* in increasing order (src/PAlgebra.cpp:532-538), returns phi(m) */
long ho_zmstar(uint64_t m, uint32_t* rep_out, long cap);
/* Phi_m(X) integer coefficients, out has phi(m)+1 entries */
void ho_phimx(uint64_t m, int64_t* out);

/* ---- per-prime transform: Cmodulus (src/CModulus.cpp) ---- */
typedef struct ho_cmod ho_cmod;
/* root: pow2 m -> w0 (primitive m-th root; 0 => FindPrimRootT(q,m));
 *       general m -> FindPrimitiveRoot output of order 2m (m even) / m (m odd);
 *       0 => computed by ho_find_prim_root (src/CModulus.cpp:148-164). */
ho_cmod* ho_cmod_create(uint64_t m, uint64_t q, uint64_t root);
void ho_cmod_destroy(ho_cmod* c);
uint64_t ho_cmod_root(const ho_cmod* c);
long ho_cmod_phim(const ho_cmod* c);
/* Cmodulus::FFT (src/CModulus.cpp:358-484): x = phim coefficients in [0,q) */
void ho_cmod_fft(const ho_cmod* c, const uint64_t* x, uint64_t* y);
/* Cmodulus::iFFT (src/CModulus.cpp:486-578) */
void ho_cmod_ifft(const ho_cmod* c, const uint64_t* y, uint64_t* x);
/* Definition of truth: y[j] = sum_i x_i zeta^(i t_j) by O(N^2) evaluation,
 * restricted to columns j in [j0, j1). */
void ho_cmod_eval_naive(const ho_cmod* c, const uint64_t* x, uint64_t* y,
                        long j0, long j1);

/* ---- DoubleCRT element-wise rows (src/DoubleCRT.cpp:135-384) ---- */
void ho_row_add(uint64_t* r, const uint64_t* a, const uint64_t* b, long n,
                uint64_t q);
void ho_row_sub(uint64_t* r, const uint64_t* a, const uint64_t* b, long n,
                uint64_t q);
void ho_row_mul(uint64_t* r, const uint64_t* a, const uint64_t* b, long n,
                uint64_t q);
void ho_row_neg(uint64_t* r, const uint64_t* a, long n, uint64_t q);
void ho_row_add_scalar(uint64_t* r, const uint64_t* a, uint64_t s, long n,
                       uint64_t q);
void ho_row_sub_scalar(uint64_t* r, const uint64_t* a, uint64_t s, long n,
                       uint64_t q);
void ho_row_mul_scalar(uint64_t* r, const uint64_t* a, uint64_t s, long n,
                       uint64_t q);
/* DoubleCRT::automorph (src/DoubleCRT.cpp:1160-1202); zms = Z_m^* reps.
 * returns -1 when k is not in Z_m^* (reference throws RuntimeError) */
int ho_row_automorph(uint64_t* out, const uint64_t* in, uint64_t m,
                     const uint32_t* zms, long phim, uint64_t k);

/* ---- DoubleCRT context: a prime chain + digit partition ---- */
typedef struct ho_ctx ho_ctx;
ho_ctx* ho_ctx_create(uint64_t m);
void ho_ctx_destroy(ho_ctx* c);
/* returns the index of the new prime (Context::moduli order) */
int ho_ctx_add_prime(ho_ctx* c, uint64_t q, uint64_t root);
long ho_ctx_phim(const ho_ctx* c);
uint64_t ho_ctx_prime(const ho_ctx* c, int idx);
uint64_t ho_ctx_root(const ho_ctx* c, int idx);
const uint32_t* ho_ctx_zms(const ho_ctx* c);

/* DoubleCRT::FFT / iFFT over a set of rows. rows[r*phim + j], prime_idx[r] */
void ho_dcrt_fft(const ho_ctx* c, const int* prime_idx, int nrows,
                 const uint64_t* coef, uint64_t* eval);
void ho_dcrt_ifft(const ho_ctx* c, const int* prime_idx, int nrows,
                  const uint64_t* eval, uint64_t* coef);

/* DoubleCRT::addPrimes (src/DoubleCRT.cpp:565-599) =
 *   toPoly (centred CRT, :925-1113) then FFT on the new primes (:68-85).
 * in : eval rows on primes from_idx[nfrom]
 * out: eval rows on primes to_idx[nto]  (the *added* rows only)
 * poly_f (optional, may be NULL): the centred coefficients as doubles
 *   (what embeddingLargestCoeff consumes, src/DoubleCRT.cpp:538-545). */
void ho_dcrt_add_primes(const ho_ctx* c, const int* from_idx, int nfrom,
                        const uint64_t* from_rows, const int* to_idx, int nto,
                        uint64_t* to_rows, double* poly_f);

/* DoubleCRT::addPrimesAndScale (src/DoubleCRT.cpp:603-647):
 * rows on from_idx are multiplied in place by prod(primes in add_idx) mod q_i.
 * (the added rows are all-zero and not materialised here) */
void ho_dcrt_scale_by_primes(const ho_ctx* c, const int* from_idx, int nfrom,
                             uint64_t* rows, const int* add_idx, int nadd);

/* DoubleCRT::breakIntoDigits (src/DoubleCRT.cpp:479-561).
 * in rows on own_idx[nown] (ctxt primes only); digit d is the subset of
 * own_idx listed in dig_idx[dig_off[d] .. dig_off[d+1]) ; all_idx[nall] =
 * own ∪ special, the order of the output rows.
 * out: digits[d][r][j] for r over all_idx, d < ndig (contiguous).       */
void ho_dcrt_break_into_digits(const ho_ctx* c, const int* own_idx, int nown,
                               const uint64_t* rows, const int* dig_idx,
                               const int* dig_off, int ndig,
                               const int* all_idx, int nall,
                               uint64_t* digits);

/* DoubleCRT::scaleDownToSet (src/DoubleCRT.cpp:1464-1516).
 * in rows on own_idx; drop_idx ⊂ own_idx are removed.  out rows on the kept
 * primes in the order they appear in own_idx.  fdelta (optional) receives
 * delta[j]/diffProd as double (src/Ctxt.cpp:466-478). */
void ho_dcrt_scale_down(const ho_ctx* c, const int* own_idx, int nown,
                        const uint64_t* rows, const int* drop_idx, int ndrop,
                        uint64_t ptxt_space, uint64_t* out_rows,
                        double* fdelta);

/* Ctxt::tensorProduct inner loop (src/Ctxt.cpp:1576-1597), 2x2 parts:
 * out = (c0*d0, c0*d1 + c1*d0, c1*d1), all rows on idx[nrows]. */
void ho_tensor(const ho_ctx* c, const int* idx, int nrows, const uint64_t* c0,
               const uint64_t* c1, const uint64_t* d0, const uint64_t* d1,
               uint64_t* o0, uint64_t* o1, uint64_t* o2);

/* Ctxt::keySwitchDigits (src/Ctxt.cpp:191-230) with explicit (a_d, b_d):
 * out0 += sum_d digit_d * b_d ; out1 += sum_d digit_d * a_d
 * all operands on all_idx[nall] rows; digits/ksk_a/ksk_b are [ndig][nall][N] */
void ho_key_switch_digits(const ho_ctx* c, const int* all_idx, int nall,
                          int ndig, const uint64_t* digits,
                          const uint64_t* ksk_b, const uint64_t* ksk_a,
                          uint64_t* out0, uint64_t* out1);

/* Full fixed-level Ctxt::multiplyBy data path:
 * tensorProduct (src/Ctxt.cpp:1563-1608) + reLinearize (src/Ctxt.cpp:720-786)
 * on prime set own_idx (ctxt primes), special primes sp_idx.
 * Output parts (2) on all_idx = own_idx followed by sp_idx.               */
void ho_mul_relin(const ho_ctx* c, const int* own_idx, int nown,
                  const int* sp_idx, int nsp, const int* dig_idx,
                  const int* dig_off, int ndig, const uint64_t* c0,
                  const uint64_t* c1, const uint64_t* d0, const uint64_t* d1,
                  const uint64_t* ksk_b, const uint64_t* ksk_a, uint64_t* out0,
                  uint64_t* out1);

/* toPoly to a centred big integer, exported for the python big-int cross
 * check: mag has nlimbs 64-bit limbs per coefficient, sign[j] in {-1,0,1}. */
int ho_dcrt_to_poly_limbs(const ho_ctx* c, const int* idx, int nrows,
                          const uint64_t* eval_rows, int positive,
                          uint64_t* mag, int nlimbs, int8_t* sign);

/* ---- canonical-embedding norm ----
 * embeddingLargestCoeff (src/norms.cpp:480-493 -> basic_/half_/quarter_ variants :129-262):
 * max over j in Z_m^*, 1 <= j <= m/2, of |sum_i f_i W^(ij)|, W = exp(2 pi i/m).  The reference
 * evaluates it with PGFFT (a complex-double DFT); here it is the definition, in long double
 * (an FFT for m a power of two, the direct sum otherwise).  Floating point: compare to 1e-9. */
double ho_embedding_largest_coeff(uint64_t m, const double* f, long n);

/* DoubleCRT::breakIntoDigits with the pieces of its return value (src/DoubleCRT.cpp:538-545):
 * frac_norms[d] = embeddingLargestCoeff(digit_d) / P_d, P_d = product of digit d's primes
 * (scaled so that it fits a double whatever the digit size).  frac_norms may be NULL. */
void ho_dcrt_break_into_digits_norms(const ho_ctx* c, const int* own_idx, int nown,
                                     const uint64_t* rows, const int* dig_idx,
                                     const int* dig_off, int ndig,
                                     const int* all_idx, int nall,
                                     uint64_t* digits, double* frac_norms);

/* deterministic test data: splitmix64 stream, rejection-sampled into [0,q) */
void ho_fill_uniform(uint64_t* out, long n, uint64_t q, uint64_t seed);

/* DoubleCRT::randomize (src/DoubleCRT.cpp:1258-1378) over a ChaCha20 (RFC 8439) stream per row;
 * see hx_oracle.c.  ho_randomize_row returns the number of 2048-byte buffers consumed. */
void ho_chacha20_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint8_t out[64]);
long ho_randomize_row(uint64_t* row, long phim, uint64_t q, const uint32_t key[8], const uint32_t nonce[3]);

#ifdef __cplusplus
}
#endif
#endif

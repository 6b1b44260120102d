/*
 * hx_oracle.c -- CPU restatement of the HElib 2.2.0 DoubleCRT hot path.
 * TEST INFRASTRUCTURE ONLY (see hx_oracle.h).  Plain C11 + unsigned __int128.
 * All file:line citations are into the HElib 2.2.0 source tree.
 */
#include "hx_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ */
/* modular primitives: NTL::MulMod / PowerMod / InvMod return the       */
/* canonical residue; any exact method reproduces them.    */
/* ------------------------------------------------------------------ */
uint64_t ho_mulmod(uint64_t a, uint64_t b, uint64_t q)
{
  return (uint64_t)(((u128)a * b) % q);
}

uint64_t ho_powmod(uint64_t a, uint64_t e, uint64_t q)
{
  uint64_t r = 1 % q;
  a %= q;
  while (e) {
    if (e & 1)
      r = ho_mulmod(r, a, q);
    a = ho_mulmod(a, a, q);
    e >>= 1;
  }
  return r;
}

uint64_t ho_invmod(uint64_t a, uint64_t q)
{
  /* extended Euclid on signed 128-bit */
  __int128 t = 0, newt = 1, r = (__int128)q, newr = (__int128)(a % q);
  while (newr != 0) {
    __int128 quo = r / newr;
    __int128 tmp = t - quo * newt;
    t = newt;
    newt = tmp;
    tmp = r - quo * newr;
    r = newr;
    newr = tmp;
  }
  if (r != 1)
    return 0; /* not invertible */
  if (t < 0)
    t += q;
  return (uint64_t)t;
}

/* Deterministic Miller-Rabin for n < 2^64 (first 12 prime bases).  Replaces
 * NTL::ProbPrime(cand, 60) at src/PrimeGenerator.h:121. */
int ho_is_prime(uint64_t n)
{
  static const uint64_t bases[12] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if (n < 2)
    return 0;
  for (int i = 0; i < 12; i++) {
    if (n == bases[i])
      return 1;
    if (n % bases[i] == 0)
      return 0;
  }
  uint64_t d = n - 1;
  int s = 0;
  while ((d & 1) == 0) {
    d >>= 1;
    s++;
  }
  for (int i = 0; i < 12; i++) {
    uint64_t x = ho_powmod(bases[i], d, n);
    if (x == 1 || x == n - 1)
      continue;
    int comp = 1;
    for (int r = 1; r < s; r++) {
      x = ho_mulmod(x, x, n);
      if (x == n - 1) {
        comp = 0;
        break;
      }
    }
    if (comp)
      return 0;
  }
  return 1;
}

/* ------------------------------------------------------------------ */
/* PrimeGenerator  (src/PrimeGenerator.h:41-126)                        */
/* ------------------------------------------------------------------ */
static long divc(long a, long b) { return (a + b - 1) / b; } /* NumbTh.h divc */

void ho_primegen_init(ho_primegen* g, long len, long m)
{
  const long B = 3;
  g->len = len;
  g->m = m;
  g->k = 0;
  while ((m << g->k) <= (1L << (len - B)))
    g->k++;
  g->t = divc((1L << len) - 1, m << g->k);
}

long ho_primegen_next(ho_primegen* g)
{
  const long B = 3;
  long len = g->len, m = g->m;
  long t_upper_bound = divc((1L << len) - 1, m << g->k);
  for (;;) {
    g->t++;
    if (g->t >= t_upper_bound) {
      g->k--;
      long k_lower_bound = (m % 2 == 0) ? 0 : 1;
      if (g->k < k_lower_bound)
        return 0; /* "Prime generator ran out of primes" */
      g->t = divc((1L << len) - (1L << (len - B)) - 1, m << g->k);
      t_upper_bound = divc((1L << len) - 1, m << g->k);
    }
    if (g->t % 2 == 0)
      continue;
    long cand = ((g->t * m) << g->k) + 1;
    if (ho_is_prime((uint64_t)cand))
      return cand;
  }
}

/* ------------------------------------------------------------------ */
/* FindPrimRootT  (src/NumbTh.cpp:436-493)                              */
/* ------------------------------------------------------------------ */
static int next_small_prime(int p)
{
  for (int c = p + 1;; c++) {
    int ok = 1;
    for (int d = 2; d * d <= c; d++)
      if (c % d == 0) {
        ok = 0;
        break;
      }
    if (ok)
      return c;
  }
}

uint64_t ho_find_prim_root(uint64_t q, uint64_t e)
{
  uint64_t qm1 = q - 1;
  if (qm1 % e != 0)
    return 0;
  /* factorize e (ascending prime factors) */
  uint64_t facts[64];
  int nf = 0;
  uint64_t ee0 = e;
  for (uint64_t p = 2; p * p <= ee0; p++) {
    if (ee0 % p == 0) {
      facts[nf++] = p;
      while (ee0 % p == 0)
        ee0 /= p;
    }
  }
  if (ee0 > 1)
    facts[nf++] = ee0;

  uint64_t root = 1;
  for (int i = 0; i < nf; i++) {
    uint64_t p = facts[i], pp = p, ee = e / p;
    while (ee % p == 0) {
      ee /= p;
      pp *= p;
    }
    int s = 1; /* NTL::PrimeSeq: 2,3,5,... */
    uint64_t qq1;
    do {
      s = next_small_prime(s);
      qq1 = ho_powmod((uint64_t)s, qm1 / p, q);
    } while (qq1 == 1);
    qq1 = ho_powmod((uint64_t)s, qm1 / pp, q);
    root = ho_mulmod(root, qq1, q);
  }
  return root;
}

/* ---- Intel HEXL (absent third-party dependency, >= 1.2.1): restated from its published
 * reference implementation; see hx_oracle.h.  Plain loops, canonical residues throughout. */
uint64_t ho_hexl_minimal_primitive_root(uint64_t q, uint64_t e)
{
  /* MinimalPrimitiveRoot(degree = e, modulus): start from any primitive e-th root, walk its odd
   * powers (root * (root^2)^i, i < e/2 -- HEXL loops to `degree`, revisiting each once), keep the
   * minimum */
  uint64_t root = ho_find_prim_root(q, e);
  if (!root)
    return 0;
  uint64_t sq = ho_mulmod(root, root, q), cur = root, best = root;
  for (uint64_t i = 0; i < e; i++) {
    if (cur < best)
      best = cur;
    cur = ho_mulmod(cur, sq, q);
  }
  return best;
}
static unsigned hexl_brev(unsigned x, int bits)
{
  unsigned r = 0;
  for (int i = 0; i < bits; i++)
    r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}
/* RootOfUnityPowers: table[brev(i)] = psi^i (NTT::ComputeRootOfUnityPowers) */
static uint64_t* hexl_root_powers(long n, uint64_t q, uint64_t psi)
{
  int bits = 0;
  while ((1L << bits) < n)
    bits++;
  uint64_t* t = (uint64_t*)malloc((size_t)n * sizeof(uint64_t));
  uint64_t cur = 1;
  for (long i = 0; i < n; i++) {
    t[hexl_brev((unsigned)i, bits)] = cur;
    cur = ho_mulmod(cur, psi, q);
  }
  return t;
}
void ho_hexl_forward(uint64_t* out, const uint64_t* in, long n, uint64_t q)
{
  const uint64_t psi = ho_hexl_minimal_primitive_root(q, (uint64_t)(2 * n));
  uint64_t* W = hexl_root_powers(n, q, psi);
  if (out != in)
    memcpy(out, in, (size_t)n * sizeof(uint64_t));
  long t = n >> 1;
  for (long m = 1; m < n; m <<= 1) {
    long j1 = 0;
    for (long i = 0; i < m; i++) {
      const uint64_t w = W[m + i];
      for (long j = j1; j < j1 + t; j++) {
        const uint64_t x = out[j], wy = ho_mulmod(out[j + t], w, q);
        out[j] = (x + wy) % q;
        out[j + t] = (x + q - wy) % q;
      }
      j1 += 2 * t;
    }
    t >>= 1;
  }
  free(W);
}
void ho_hexl_inverse(uint64_t* out, const uint64_t* in, long n, uint64_t q)
{
  const uint64_t psi = ho_hexl_minimal_primitive_root(q, (uint64_t)(2 * n));
  /* InvRootOfUnityPowers, consumed by the Gentleman-Sande network stage by stage: here indexed
   * as the forward table of psi^-1 (same values the network needs at (m + i)) */
  uint64_t* W = hexl_root_powers(n, q, ho_invmod(psi, q));
  if (out != in)
    memcpy(out, in, (size_t)n * sizeof(uint64_t));
  long t = 1;
  for (long m = n >> 1; m >= 1; m >>= 1) {
    long j1 = 0;
    for (long i = 0; i < m; i++) {
      const uint64_t w = W[m + i];
      for (long j = j1; j < j1 + t; j++) {
        const uint64_t x = out[j], y = out[j + t];
        out[j] = (x + y) % q;
        out[j + t] = ho_mulmod((x + q - y) % q, w, q);
      }
      j1 += 2 * t;
    }
    t <<= 1;
  }
  const uint64_t ninv = ho_invmod((uint64_t)n % q, q);
  for (long j = 0; j < n; j++)
    out[j] = ho_mulmod(out[j], ninv, q);
  free(W);
}

/* ------------------------------------------------------------------ */
/* Z_m^* and Phi_m                                                       */
/* ------------------------------------------------------------------ */
static uint64_t gcd_u64(uint64_t a, uint64_t b)
{
  while (b) {
    uint64_t t = a % b;
    a = b;
    b = t;
  }
  return a;
}

long ho_zmstar(uint64_t m, uint32_t* rep_out, long cap)
{
  long n = 0;
  for (uint64_t i = 1; i < m || (m == 1 && i == 1); i++) {
    if (gcd_u64(i, m) == 1) {
      if (rep_out && n < cap)
        rep_out[n] = (uint32_t)i;
      n++;
    }
    if (m == 1)
      break;
  }
  return n;
}

/* Phi_m(X) = prod_{d|m} (X^d - 1)^{mu(m/d)} computed by exact integer
 * multiplication/division of sparse binomials. */
static int mobius_i(uint64_t n)
{
  int mu = 1;
  for (uint64_t p = 2; p * p <= n; p++) {
    if (n % p == 0) {
      n /= p;
      if (n % p == 0)
        return 0;
      mu = -mu;
    }
  }
  if (n > 1)
    mu = -mu;
  return mu;
}

void ho_phimx(uint64_t m, int64_t* out)
{
  long phi = ho_zmstar(m, NULL, 0);
  /* work modulo X^(phi+1): numerator/denominator products of (X^d - 1) */
  long L = phi + 1;
  int64_t* a = (int64_t*)calloc((size_t)L, sizeof(int64_t));
  a[0] = 1;
  /* multiply by (X^d - 1) for mu=+1 (truncated) */
  for (uint64_t d = 1; d <= m; d++) {
    if (m % d)
      continue;
    if (mobius_i(m / d) == 1) {
      for (long i = L - 1; i >= 0; i--) {
        int64_t v = -a[i];
        if (i >= (long)d)
          v += a[i - (long)d];
        a[i] = v;
      }
    }
  }
  /* divide by (X^d - 1) for mu=-1: power series division (truncated)
   * a / (X^d - 1) = -a / (1 - X^d)  => b[i] = -a[i] + b[i-d] */
  for (uint64_t d = 1; d <= m; d++) {
    if (m % d)
      continue;
    if (mobius_i(m / d) == -1) {
      for (long i = 0; i < L; i++) {
        int64_t v = -a[i];
        if (i >= (long)d)
          v += a[i - (long)d];
        a[i] = v;
      }
    }
  }
  memcpy(out, a, (size_t)L * sizeof(int64_t));
  free(a);
}

/* ------------------------------------------------------------------ */
/* plain cyclic radix-2 NTT of length n=2^lg, natural order in and out  */
/* (restates "NTL::FFTFwd + BitReverseCopy": src/CModulus.cpp:408-426)  */
/* ------------------------------------------------------------------ */
static void bitrev_permute(uint64_t* a, long n)
{
  for (long i = 1, j = 0; i < n; i++) {
    long bit = n >> 1;
    for (; j & bit; bit >>= 1)
      j ^= bit;
    j ^= bit;
    if (i < j) {
      uint64_t t = a[i];
      a[i] = a[j];
      a[j] = t;
    }
  }
}

/* x*w mod q (canonical) with wp = floor(w 2^64 / q), any x < 2^64, q < 2^63: Shoup's form of the
 * same product ho_mulmod computes -- r = x w - floor(x wp / 2^64) q lies in [0, 2q).  Used for the
 * butterflies only (what NTL's MulModPrecon does in its FFT), so that the CPU baseline bench.py
 * reports is not dominated by 128-by-64-bit divisions; every value is the same canonical residue. */
static inline uint64_t mulmod_precon(uint64_t x, uint64_t w, uint64_t wp, uint64_t q)
{
  uint64_t h = (uint64_t)(((u128)x * wp) >> 64);
  uint64_t r = x * w - h * q;
  return r - (q & (0 - (uint64_t)(r >= q)));   /* (branch-free: the outcome is a coin flip on random data) */
}

/* The stage tables (w^j and their Shoup companions for every stage, n - 1 entries each) are built
 * once per (q, n, omega) and kept per thread -- NTL keeps its FFT tables per prime the same way
 * (FFTTablesType); building them by a dependent chain of ho_mulmod on every call cost more than
 * the butterflies. */
typedef struct {
  uint64_t q, omega;
  long n;
  uint64_t* w;  /* [2*(n-1)]: stage with half-length h starts at h-1; wp after the w block */
} ntt_tab;
static __thread ntt_tab* g_tabs = NULL;
static __thread long g_ntabs = 0, g_tabcap = 0;

static const uint64_t* ntt_tables(long n, uint64_t omega, uint64_t q)
{
  for (long i = 0; i < g_ntabs; i++)
    if (g_tabs[i].q == q && g_tabs[i].n == n && g_tabs[i].omega == omega)
      return g_tabs[i].w;
  if (g_ntabs == g_tabcap) {
    g_tabcap = g_tabcap ? 2 * g_tabcap : 64;
    g_tabs = (ntt_tab*)realloc(g_tabs, (size_t)g_tabcap * sizeof(ntt_tab));
  }
  uint64_t* w = (uint64_t*)malloc((size_t)2 * (size_t)n * sizeof(uint64_t));
  uint64_t* wp = w + n;
  for (long len = 2; len <= n; len <<= 1) {
    uint64_t wlen = ho_powmod(omega, (uint64_t)(n / len), q);
    long half = len >> 1;
    uint64_t* ws = w + (half - 1);
    ws[0] = 1 % q;
    for (long j = 1; j < half; j++)
      ws[j] = ho_mulmod(ws[j - 1], wlen, q);
    for (long j = 0; j < half; j++)
      wp[half - 1 + j] = (uint64_t)((((u128)ws[j]) << 64) / q);
  }
  ntt_tab t = {q, omega, n, w};
  g_tabs[g_ntabs++] = t;
  return w;
}

static void cyclic_ntt(uint64_t* a, long n, uint64_t omega, uint64_t q)
{
  bitrev_permute(a, n);
  const uint64_t* tab = ntt_tables(n, omega, q);
  for (long len = 2; len <= n; len <<= 1) {
    long half = len >> 1;
    const uint64_t* w = tab + (half - 1);
    const uint64_t* wp = tab + n + (half - 1);
    for (long i = 0; i < n; i += len) {
      for (long j = 0; j < half; j++) {
        uint64_t u = a[i + j];
        uint64_t v = mulmod_precon(a[i + j + half], w[j], wp[j], q);
        uint64_t s = u + v;
        s -= q & (0 - (uint64_t)(s >= q));
        uint64_t d = u - v + (q & (0 - (uint64_t)(u < v)));
        a[i + j] = s;
        a[i + j + half] = d;
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* Cmodulus                                                             */
/* ------------------------------------------------------------------ */
struct ho_cmod {
  uint64_t m, q;
  long phim;
  int pow2k; /* k if m == 2^k, else 0   (PAlgebra::getPow2) */
  uint64_t root, rInv, m_inv;
  uint32_t* zms; /* Z_m^* reps, increasing */
  /* pow2: powers[i] = w0^i, ipowers[i] = w1^i  (src/CModulus.cpp:118-135) */
  uint64_t *powers, *ipowers;
  /* pow2 only: Shoup companions of powers[i], and ipowers[i]/N with its companions (the
   * reference's MulModPrecon tables, src/CModulus.cpp:121-135: same canonical products) */
  uint64_t *powers_p, *ipn, *ipn_p;
  /* Bluestein (src/bluestein.cpp:76-132): powers[i]=root^{i^2}, chirp b */
  long bk, bk2;      /* conv length 2^bk */
  uint64_t *b, *ib;  /* chirp polys, length bk2 (time domain) */
  uint64_t conv_w;   /* primitive 2^bk-th root of unity mod q, or 0 */
  uint64_t *Rb, *iRb; /* NTT of chirps when conv_w != 0 */
  int64_t* phimx;    /* Phi_m(X), phim+1 coeffs */
};

static long next_pow2_exp(long n) /* NTL::NextPowerOfTwo: least k, 2^k >= n */
{
  long k = 0;
  while ((1L << k) < n)
    k++;
  return k;
}

/* BluesteinInit (src/bluestein.cpp:76-132) */
static void bluestein_init(const ho_cmod* c, uint64_t root, uint64_t* powers,
                           uint64_t* b)
{
  long n = (long)c->m;
  uint64_t q = c->q;
  uint64_t e = (n % 2 == 0) ? 2 * (uint64_t)n : (uint64_t)n;
  powers[0] = 1;
  for (long i = 1; i < n; i++) {
    uint64_t iSqr = ho_mulmod((uint64_t)i, (uint64_t)i, e);
    powers[i] = ho_powmod(root, iSqr, q);
  }
  memset(b, 0, (size_t)c->bk2 * sizeof(uint64_t));
  uint64_t rInv = ho_invmod(root, q);
  if ((uint64_t)n == e) { /* NEW_BLUE && n odd */
    for (long i = 0; i < n; i++) {
      uint64_t iSqr = ho_mulmod((uint64_t)i, (uint64_t)i, e);
      b[i] = ho_powmod(rInv, iSqr, q);
    }
  } else {
    b[n - 1] = 1;
    for (long i = 1; i < n; i++) {
      uint64_t iSqr = ho_mulmod((uint64_t)i, (uint64_t)i, e);
      uint64_t bi = ho_powmod(rInv, iSqr, q);
      b[n - 1 + i] = bi;
      b[n - 1 - i] = bi;
    }
  }
}

ho_cmod* ho_cmod_create(uint64_t m, uint64_t q, uint64_t root)
{
  ho_cmod* c = (ho_cmod*)calloc(1, sizeof(ho_cmod));
  c->m = m;
  c->q = q;
  c->phim = ho_zmstar(m, NULL, 0);
  c->zms = (uint32_t*)malloc((size_t)c->phim * sizeof(uint32_t));
  ho_zmstar(m, c->zms, c->phim);
  c->m_inv = ho_invmod(m % q, q);
  c->pow2k = 0;
  if (m >= 2 && (m & (m - 1)) == 0) {
    int k = 0;
    while ((1ULL << k) < m)
      k++;
    c->pow2k = k;
  }
  if (c->pow2k) {
    /* src/CModulus.cpp:84-138.  w0 = RootTable[0][k] is an INPUT (NTL PRG);
     * fallback FindPrimRootT(q, m): "parity unpinned" vs an NTL build. */
    if (root == 0)
      root = ho_find_prim_root(q, m);
    c->root = root;
    c->rInv = ho_invmod(root, q);
    long phim = c->phim;
    c->powers = (uint64_t*)malloc((size_t)phim * 8);
    c->ipowers = (uint64_t*)malloc((size_t)phim * 8);
    c->powers_p = (uint64_t*)malloc((size_t)phim * 8);
    c->ipn = (uint64_t*)malloc((size_t)phim * 8);
    c->ipn_p = (uint64_t*)malloc((size_t)phim * 8);
    const uint64_t ninv = ho_invmod((uint64_t)phim % q, q);
    uint64_t w = 1, iw = 1;
    for (long i = 0; i < phim; i++) {
      c->powers[i] = w;
      c->ipowers[i] = iw;
      c->powers_p[i] = (uint64_t)((((u128)w) << 64) / q);
      c->ipn[i] = ho_mulmod(iw, ninv, q);
      c->ipn_p[i] = (uint64_t)((((u128)c->ipn[i]) << 64) / q);
      w = ho_mulmod(w, c->root, q);
      iw = ho_mulmod(iw, c->rInv, q);
    }
    return c;
  }
  /* general m: src/CModulus.cpp:140-181 */
  if (root == 0) {
    uint64_t e = (m % 2 == 0) ? 2 * m : m;
    root = ho_find_prim_root(q, e);
  }
  c->root = root;
  c->rInv = ho_invmod(root, q);
  c->bk = next_pow2_exp(2 * (long)m - 1);
  c->bk2 = 1L << c->bk;
  c->powers = (uint64_t*)malloc((size_t)m * 8);
  c->ipowers = (uint64_t*)malloc((size_t)m * 8);
  c->b = (uint64_t*)malloc((size_t)c->bk2 * 8);
  c->ib = (uint64_t*)malloc((size_t)c->bk2 * 8);
  bluestein_init(c, c->root, c->powers, c->b);
  bluestein_init(c, c->rInv, c->ipowers, c->ib);
  /* NTL's fftRep convolution is exact mod q whatever its internals; we use
   * an NTT mod q when 2^bk | q-1 and schoolbook otherwise. */
  c->conv_w = 0;
  if (((q - 1) % (uint64_t)c->bk2) == 0) {
    c->conv_w = ho_find_prim_root(q, (uint64_t)c->bk2);
    c->Rb = (uint64_t*)malloc((size_t)c->bk2 * 8);
    c->iRb = (uint64_t*)malloc((size_t)c->bk2 * 8);
    memcpy(c->Rb, c->b, (size_t)c->bk2 * 8);
    memcpy(c->iRb, c->ib, (size_t)c->bk2 * 8);
    cyclic_ntt(c->Rb, c->bk2, c->conv_w, q);
    cyclic_ntt(c->iRb, c->bk2, c->conv_w, q);
  }
  c->phimx = (int64_t*)malloc((size_t)(c->phim + 1) * sizeof(int64_t));
  ho_phimx(m, c->phimx);
  return c;
}

void ho_cmod_destroy(ho_cmod* c)
{
  if (!c)
    return;
  free(c->zms);
  free(c->powers);
  free(c->ipowers);
  free(c->powers_p);
  free(c->ipn);
  free(c->ipn_p);
  free(c->b);
  free(c->ib);
  free(c->Rb);
  free(c->iRb);
  free(c->phimx);
  free(c);
}

uint64_t ho_cmod_root(const ho_cmod* c) { return c->root; }
long ho_cmod_phim(const ho_cmod* c) { return c->phim; }

/* BluesteinFFT (src/bluestein.cpp:134-201). x has m entries (deg < m),
 * transformed in place. */
static void bluestein_fft(const ho_cmod* c, uint64_t* x, const uint64_t* powers,
                          const uint64_t* b, const uint64_t* Rb)
{
  long n = (long)c->m;
  uint64_t q = c->q;
  int allzero = 1;
  for (long i = 0; i < n; i++)
    if (x[i]) {
      allzero = 0;
      break;
    }
  if (allzero)
    return; /* :143 IsZero(x) */
  for (long i = 0; i < n; i++)
    x[i] = ho_mulmod(x[i], powers[i], q); /* :153-156 */

  long k2 = c->bk2;
  uint64_t* h = (uint64_t*)calloc((size_t)k2, 8);
  if (c->conv_w) {
    memcpy(h, x, (size_t)n * 8);
    cyclic_ntt(h, k2, c->conv_w, q);
    for (long i = 0; i < k2; i++)
      h[i] = ho_mulmod(h[i], Rb[i], q);
    /* inverse cyclic NTT */
    cyclic_ntt(h, k2, ho_invmod(c->conv_w, q), q);
    uint64_t k2inv = ho_invmod((uint64_t)k2 % q, q);
    for (long i = 0; i < k2; i++)
      h[i] = ho_mulmod(h[i], k2inv, q);
  } else {
    /* schoolbook cyclic convolution mod X^k2 - 1 (small sizes only) */
    for (long i = 0; i < n; i++) {
      if (!x[i])
        continue;
      for (long j = 0; j < k2; j++) {
        if (!b[j])
          continue;
        long idx = (i + j) & (k2 - 1);
        uint64_t t = ho_mulmod(x[i], b[j], q);
        uint64_t s = h[idx] + t;
        if (s >= q)
          s -= q;
        h[idx] = s;
      }
    }
  }
  if (n % 2 != 0) {
    /* :166-187 : coefficients 0..2(n-1), fold mod x^n - 1, twist */
    for (long i = n; i <= 2 * (n - 1); i++) {
      uint64_t s = h[i - n] + h[i];
      if (s >= q)
        s -= q;
      h[i - n] = s;
    }
    for (long i = 0; i < n; i++)
      x[i] = ho_mulmod(h[i], powers[i], q);
  } else {
    /* :189-199 : window [n-1, 2n-2] */
    for (long i = 0; i < n; i++)
      x[i] = ho_mulmod(h[n - 1 + i], powers[i], q);
  }
  free(h);
}

void ho_cmod_fft(const ho_cmod* c, const uint64_t* x, uint64_t* y)
{
  uint64_t q = c->q;
  long phim = c->phim;
  if (c->pow2k) {
    /* src/CModulus.cpp:389-426 */
    uint64_t* t = (uint64_t*)malloc((size_t)phim * 8);
    for (long i = 0; i < phim; i++)
      t[i] = mulmod_precon(x[i], c->powers[i], c->powers_p[i], q);   /* (any x < 2^64: the % q is implied) */
    cyclic_ntt(t, phim, ho_mulmod(c->root, c->root, q), q);
    memcpy(y, t, (size_t)phim * 8);
    free(t);
    return;
  }
  /* src/CModulus.cpp:431-443 */
  long m = (long)c->m;
  uint64_t* t = (uint64_t*)calloc((size_t)m, 8);
  for (long i = 0; i < phim; i++)
    t[i] = x[i] % q;
  bluestein_fft(c, t, c->powers, c->b, c->Rb);
  for (long j = 0; j < phim; j++)
    y[j] = t[c->zms[j]];
  free(t);
}

void ho_cmod_ifft(const ho_cmod* c, const uint64_t* y, uint64_t* x)
{
  uint64_t q = c->q;
  long phim = c->phim;
  if (c->pow2k) {
    /* src/CModulus.cpp:493-553: bitrev, FFTRev1 (incl 1/N), untwist */
    uint64_t* t = (uint64_t*)malloc((size_t)phim * 8);
    memcpy(t, y, (size_t)phim * 8);
    uint64_t w1sq = ho_mulmod(c->rInv, c->rInv, q);
    cyclic_ntt(t, phim, w1sq, q);
    for (long i = 0; i < phim; i++)   /* (t/N) * w1^i as one product with ipn[i] = w1^i / N */
      x[i] = mulmod_precon(t[i], c->ipn[i], c->ipn_p[i], q);
    free(t);
    return;
  }
  /* src/CModulus.cpp:555-577 */
  long m = (long)c->m;
  uint64_t* t = (uint64_t*)calloc((size_t)m, 8);
  for (long j = 0; j < phim; j++)
    t[c->zms[j]] = y[j];
  bluestein_fft(c, t, c->ipowers, c->ib, c->iRb);
  /* rem(x, Phi_m): schoolbook division by the monic Phi_m (exact) */
  for (long i = m - 1; i >= phim; i--) {
    uint64_t ci = t[i];
    if (!ci)
      continue;
    for (long j = 0; j <= phim; j++) {
      int64_t f = c->phimx[j];
      if (!f)
        continue;
      uint64_t fa = (uint64_t)(f < 0 ? -f : f) % q;
      uint64_t prod = (fa == 1) ? ci : ho_mulmod(ci, fa, q);
      long idx = i - phim + j;
      if (f > 0)
        t[idx] = t[idx] >= prod ? t[idx] - prod : t[idx] + q - prod;
      else {
        uint64_t s = t[idx] + prod;
        t[idx] = s >= q ? s - q : s;
      }
    }
  }
  for (long i = 0; i < phim; i++)
    x[i] = ho_mulmod(t[i], c->m_inv, q);
  free(t);
}

void ho_cmod_eval_naive(const ho_cmod* c, const uint64_t* x, uint64_t* y,
                        long j0, long j1)
{
  /* zeta = w0 for m=2^k, root^2 for general m
   * (include/helib/bluestein.h:23-24) */
  uint64_t q = c->q;
  uint64_t zeta = c->pow2k ? c->root : ho_mulmod(c->root, c->root, q);
  for (long j = j0; j < j1; j++) {
    uint64_t pt = ho_powmod(zeta, c->zms[j], q);
    uint64_t acc = 0;
    for (long i = c->phim - 1; i >= 0; i--) {
      acc = ho_mulmod(acc, pt, q);
      acc += x[i] % q;
      if (acc >= q)
        acc -= q;
    }
    y[j - j0] = acc;
  }
}

/* ------------------------------------------------------------------ */
/* DoubleCRT rows  (src/DoubleCRT.cpp:135-213 functors, :216-384)       */
/* ------------------------------------------------------------------ */
void ho_row_add(uint64_t* r, const uint64_t* a, const uint64_t* b, long n,
                uint64_t q)
{
  for (long j = 0; j < n; j++) {
    uint64_t s = a[j] + b[j];
    r[j] = s >= q ? s - q : s;
  }
}
void ho_row_sub(uint64_t* r, const uint64_t* a, const uint64_t* b, long n,
                uint64_t q)
{
  for (long j = 0; j < n; j++)
    r[j] = a[j] >= b[j] ? a[j] - b[j] : a[j] + q - b[j];
}
void ho_row_mul(uint64_t* r, const uint64_t* a, const uint64_t* b, long n,
                uint64_t q)
{
  for (long j = 0; j < n; j++)
    r[j] = ho_mulmod(a[j], b[j], q);
}
void ho_row_neg(uint64_t* r, const uint64_t* a, long n, uint64_t q)
{
  for (long j = 0; j < n; j++)
    r[j] = a[j] ? q - a[j] : 0;
}
void ho_row_add_scalar(uint64_t* r, const uint64_t* a, uint64_t s, long n,
                       uint64_t q)
{
  s %= q;
  for (long j = 0; j < n; j++) {
    uint64_t v = a[j] + s;
    r[j] = v >= q ? v - q : v;
  }
}
void ho_row_sub_scalar(uint64_t* r, const uint64_t* a, uint64_t s, long n,
                       uint64_t q)
{
  s %= q;
  for (long j = 0; j < n; j++)
    r[j] = a[j] >= s ? a[j] - s : a[j] + q - s;
}
void ho_row_mul_scalar(uint64_t* r, const uint64_t* a, uint64_t s, long n,
                       uint64_t q)
{
  s %= q;
  for (long j = 0; j < n; j++)
    r[j] = ho_mulmod(a[j], s, q);
}

int ho_row_automorph(uint64_t* out, const uint64_t* in, uint64_t m,
                     const uint32_t* zms, long phim, uint64_t k)
{
  /* src/DoubleCRT.cpp:1160-1202: new[j] = old[idx(t_j * k mod m)] */
  k %= m;
  if (gcd_u64(k, m) != 1)
    return -1;
  uint64_t* tmp = (uint64_t*)calloc((size_t)m, 8);
  for (long j = 0; j < phim; j++)
    tmp[zms[j]] = in[j];
  for (long j = 0; j < phim; j++)
    out[j] = tmp[ho_mulmod(zms[j], k, m)];
  free(tmp);
  return 0;
}

/* ------------------------------------------------------------------ */
/* tiny fixed-width big integers (replaces NTL::ZZ in toPoly)           */
/* little-endian 64-bit limbs, two's complement when signed             */
/* ------------------------------------------------------------------ */
static void bn_zero(uint64_t* a, int nl) { memset(a, 0, (size_t)nl * 8); }
static void bn_copy(uint64_t* r, const uint64_t* a, int nl)
{
  memcpy(r, a, (size_t)nl * 8);
}
static void bn_set_word(uint64_t* a, int nl, uint64_t w)
{
  bn_zero(a, nl);
  a[0] = w;
}
static void bn_mul_word(uint64_t* a, int nl, uint64_t w)
{
  uint64_t carry = 0;
  for (int i = 0; i < nl; i++) {
    u128 p = (u128)a[i] * w + carry;
    a[i] = (uint64_t)p;
    carry = (uint64_t)(p >> 64);
  }
}
static void bn_addmul_word(uint64_t* acc, const uint64_t* a, int nl, uint64_t w)
{
  uint64_t carry = 0;
  for (int i = 0; i < nl; i++) {
    u128 p = (u128)a[i] * w + acc[i] + carry;
    acc[i] = (uint64_t)p;
    carry = (uint64_t)(p >> 64);
  }
}
static void bn_add(uint64_t* a, const uint64_t* b, int nl)
{
  unsigned char c = 0;
  for (int i = 0; i < nl; i++) {
    u128 s = (u128)a[i] + b[i] + c;
    a[i] = (uint64_t)s;
    c = (unsigned char)(s >> 64);
  }
}
static void bn_sub(uint64_t* a, const uint64_t* b, int nl)
{
  unsigned char br = 0;
  for (int i = 0; i < nl; i++) {
    u128 d = (u128)a[i] - b[i] - br;
    a[i] = (uint64_t)d;
    br = (unsigned char)((d >> 64) & 1);
  }
}
static int bn_cmp(const uint64_t* a, const uint64_t* b, int nl) /* unsigned */
{
  for (int i = nl - 1; i >= 0; i--) {
    if (a[i] != b[i])
      return a[i] < b[i] ? -1 : 1;
  }
  return 0;
}
static int bn_is_neg(const uint64_t* a, int nl) { return (int)(a[nl - 1] >> 63); }
static int bn_is_zero(const uint64_t* a, int nl)
{
  for (int i = 0; i < nl; i++)
    if (a[i])
      return 0;
  return 1;
}
static void bn_negate(uint64_t* a, int nl)
{
  unsigned char c = 1;
  for (int i = 0; i < nl; i++) {
    u128 s = (u128)(~a[i]) + c;
    a[i] = (uint64_t)s;
    c = (unsigned char)(s >> 64);
  }
}
static uint64_t bn_mod_word(const uint64_t* a, int nl, uint64_t q) /* unsigned */
{
  u128 r = 0;
  for (int i = nl - 1; i >= 0; i--)
    r = ((r << 64) | a[i]) % q;
  return (uint64_t)r;
}
static uint64_t bn_div_word(uint64_t* a, int nl, uint64_t q) /* a /= q, ret rem */
{
  u128 r = 0;
  for (int i = nl - 1; i >= 0; i--) {
    u128 cur = (r << 64) | a[i];
    a[i] = (uint64_t)(cur / q);
    r = cur % q;
  }
  return (uint64_t)r;
}
static void bn_shr1(uint64_t* a, int nl)
{
  for (int i = 0; i < nl; i++) {
    uint64_t hi = (i + 1 < nl) ? a[i + 1] : 0;
    a[i] = (a[i] >> 1) | (hi << 63);
  }
}
/* signed (two's complement) value mod q, canonical residue */
static uint64_t bn_smod_word(const uint64_t* a, int nl, uint64_t q, uint64_t* tmp)
{
  if (!bn_is_neg(a, nl))
    return bn_mod_word(a, nl, q);
  bn_copy(tmp, a, nl);
  bn_negate(tmp, nl);
  uint64_t r = bn_mod_word(tmp, nl, q);
  return r ? q - r : 0;
}
static long double bn_to_ldouble_signed(const uint64_t* a, int nl, uint64_t* tmp)
{
  int neg = bn_is_neg(a, nl);
  const uint64_t* mag = a;
  if (neg) {
    bn_copy(tmp, a, nl);
    bn_negate(tmp, nl);
    mag = tmp;
  }
  long double v = 0;
  for (int i = nl - 1; i >= 0; i--)
    v = v * 18446744073709551616.0L + (long double)mag[i];
  return neg ? -v : v;
}
static double bn_to_double_signed(const uint64_t* a, int nl, uint64_t* tmp)
{
  int neg = bn_is_neg(a, nl);
  const uint64_t* mag = a;
  if (neg) {
    bn_copy(tmp, a, nl);
    bn_negate(tmp, nl);
    mag = tmp;
  }
  long double v = 0;
  for (int i = nl - 1; i >= 0; i--)
    v = v * 18446744073709551616.0L + (long double)mag[i];
  return neg ? -(double)v : (double)v;
}

/* ------------------------------------------------------------------ */
/* context                                                              */
/* ------------------------------------------------------------------ */
struct ho_ctx {
  uint64_t m;
  long phim;
  uint32_t* zms;
  int nprimes, cap;
  ho_cmod** mod;
};

ho_ctx* ho_ctx_create(uint64_t m)
{
  ho_ctx* c = (ho_ctx*)calloc(1, sizeof(ho_ctx));
  c->m = m;
  c->phim = ho_zmstar(m, NULL, 0);
  c->zms = (uint32_t*)malloc((size_t)c->phim * 4);
  ho_zmstar(m, c->zms, c->phim);
  c->cap = 16;
  c->mod = (ho_cmod**)calloc((size_t)c->cap, sizeof(ho_cmod*));
  return c;
}
void ho_ctx_destroy(ho_ctx* c)
{
  if (!c)
    return;
  for (int i = 0; i < c->nprimes; i++)
    ho_cmod_destroy(c->mod[i]);
  free(c->mod);
  free(c->zms);
  free(c);
}
int ho_ctx_add_prime(ho_ctx* c, uint64_t q, uint64_t root)
{
  if (c->nprimes == c->cap) {
    c->cap *= 2;
    c->mod = (ho_cmod**)realloc(c->mod, (size_t)c->cap * sizeof(ho_cmod*));
  }
  c->mod[c->nprimes] = ho_cmod_create(c->m, q, root);
  return c->nprimes++;
}
long ho_ctx_phim(const ho_ctx* c) { return c->phim; }
uint64_t ho_ctx_prime(const ho_ctx* c, int idx) { return c->mod[idx]->q; }
uint64_t ho_ctx_root(const ho_ctx* c, int idx) { return c->mod[idx]->root; }
const uint32_t* ho_ctx_zms(const ho_ctx* c) { return c->zms; }

void ho_dcrt_fft(const ho_ctx* c, const int* prime_idx, int nrows,
                 const uint64_t* coef, uint64_t* eval)
{
  long N = c->phim;
  for (int r = 0; r < nrows; r++)
    ho_cmod_fft(c->mod[prime_idx[r]], coef + (size_t)r * N, eval + (size_t)r * N);
}
void ho_dcrt_ifft(const ho_ctx* c, const int* prime_idx, int nrows,
                  const uint64_t* eval, uint64_t* coef)
{
  long N = c->phim;
  for (int r = 0; r < nrows; r++)
    ho_cmod_ifft(c->mod[prime_idx[r]], eval + (size_t)r * N, coef + (size_t)r * N);
}

/* ------------------------------------------------------------------ */
/* toPoly (src/DoubleCRT.cpp:925-1113): iFFT per prime, then integer    */
/* CRT per coefficient; result centred in [-P/2, P/2) unless positive.  */
/* Output: two's complement big ints, nl limbs per coefficient.         */
/* ------------------------------------------------------------------ */
typedef struct {
  int icard, nl;
  uint64_t* q;      /* primes                                          */
  uint64_t* tvec;   /* (prod/q_i)^{-1} mod q_i          (:1044-1051)   */
  uint64_t* prod;   /* product of the primes                           */
  uint64_t* half;   /* (prod+1)/2                        (:1056-1059)  */
  uint64_t* prod1;  /* prod/q_i, icard x nl                            */
} crt_tab;

static int limbs_for(const ho_ctx* c, const int* idx, int n, int extra_bits)
{
  double bits = extra_bits;
  for (int i = 0; i < n; i++)
    bits += log2((double)c->mod[idx[i]]->q);
  return (int)(bits / 64.0) + 3;
}

static void crt_tab_init(crt_tab* t, const ho_ctx* c, const int* idx, int n, int nl)
{
  t->icard = n;
  t->nl = nl;
  t->q = (uint64_t*)malloc((size_t)n * 8);
  t->tvec = (uint64_t*)malloc((size_t)n * 8);
  t->prod = (uint64_t*)malloc((size_t)nl * 8);
  t->half = (uint64_t*)malloc((size_t)nl * 8);
  t->prod1 = (uint64_t*)malloc((size_t)n * nl * 8);
  bn_set_word(t->prod, nl, 1);
  for (int j = 0; j < n; j++) {
    t->q[j] = c->mod[idx[j]]->q;
    bn_mul_word(t->prod, nl, t->q[j]);
  }
  for (int j = 0; j < n; j++) {
    uint64_t* p1 = t->prod1 + (size_t)j * nl;
    bn_copy(p1, t->prod, nl);
    bn_div_word(p1, nl, t->q[j]);
    uint64_t r = bn_mod_word(p1, nl, t->q[j]);
    t->tvec[j] = ho_invmod(r, t->q[j]);
  }
  bn_copy(t->half, t->prod, nl);
  uint64_t one[1] = {1};
  uint64_t* onebn = (uint64_t*)calloc((size_t)nl, 8);
  onebn[0] = one[0];
  bn_add(t->half, onebn, nl);
  bn_shr1(t->half, nl);
  free(onebn);
}
static void crt_tab_free(crt_tab* t)
{
  free(t->q);
  free(t->tvec);
  free(t->prod);
  free(t->half);
  free(t->prod1);
}

/* CRT one coefficient: rem[j] = residue mod q_j. out: nl limbs two's compl. */
static void crt_one(const crt_tab* t, const uint64_t* rem, int positive,
                    uint64_t* out)
{
  int nl = t->nl;
  bn_zero(out, nl);
  for (int j = 0; j < t->icard; j++) {
    uint64_t r = ho_mulmod(rem[j], t->tvec[j], t->q[j]); /* :1083 */
    bn_addmul_word(out, t->prod1 + (size_t)j * nl, nl, r); /* :1084 */
  }
  /* reduce modulo prod (:1087-1092); the reference subtracts a double
   * estimate of the quotient and fixes up -- same result as exact mod. */
  while (bn_cmp(out, t->prod, nl) >= 0)
    bn_sub(out, t->prod, nl);
  if (!positive && bn_cmp(out, t->half, nl) >= 0)
    bn_sub(out, t->prod, nl); /* :1098-1099 */
}

/* eval rows (nrows x N) -> per-coefficient residues remtab[h*nrows + j] */
static uint64_t* ifft_to_remtab(const ho_ctx* c, const int* idx, int nrows,
                                const uint64_t* eval)
{
  long N = c->phim;
  uint64_t* coef = (uint64_t*)malloc((size_t)N * 8);
  uint64_t* remtab = (uint64_t*)malloc((size_t)N * nrows * 8);
  for (int r = 0; r < nrows; r++) {
    ho_cmod_ifft(c->mod[idx[r]], eval + (size_t)r * N, coef);
    for (long h = 0; h < N; h++)
      remtab[(size_t)h * nrows + r] = coef[h];
  }
  free(coef);
  return remtab;
}

int ho_dcrt_to_poly_limbs(const ho_ctx* c, const int* idx, int nrows,
                          const uint64_t* eval_rows, int positive,
                          uint64_t* mag, int nlimbs, int8_t* sign)
{
  long N = c->phim;
  int nl = limbs_for(c, idx, nrows, 8);
  if (nl > nlimbs + 2)
    return -1;
  crt_tab t;
  crt_tab_init(&t, c, idx, nrows, nl);
  uint64_t* remtab = ifft_to_remtab(c, idx, nrows, eval_rows);
  uint64_t* v = (uint64_t*)malloc((size_t)nl * 8);
  for (long h = 0; h < N; h++) {
    crt_one(&t, remtab + (size_t)h * nrows, positive, v);
    int neg = bn_is_neg(v, nl);
    if (neg)
      bn_negate(v, nl);
    sign[h] = bn_is_zero(v, nl) ? 0 : (neg ? -1 : 1);
    for (int i = 0; i < nlimbs; i++)
      mag[(size_t)h * nlimbs + i] = i < nl ? v[i] : 0;
  }
  free(v);
  free(remtab);
  crt_tab_free(&t);
  return 0;
}

/* ------------------------------------------------------------------ */
/* addPrimes  (src/DoubleCRT.cpp:565-599)                               */
/* ------------------------------------------------------------------ */
static void add_primes_impl(const ho_ctx* c, const int* from_idx, int nfrom,
                            const uint64_t* from_rows, const int* to_idx, int nto,
                            uint64_t* to_rows, double* poly_f, double* poly_frac);
void ho_dcrt_add_primes(const ho_ctx* c, const int* from_idx, int nfrom,
                        const uint64_t* from_rows, const int* to_idx, int nto,
                        uint64_t* to_rows, double* poly_f)
{
  add_primes_impl(c, from_idx, nfrom, from_rows, to_idx, nto, to_rows, poly_f, NULL);
}
/* poly_frac (optional): the centred coefficients divided by the product of the from-primes */
static void add_primes_impl(const ho_ctx* c, const int* from_idx, int nfrom,
                            const uint64_t* from_rows, const int* to_idx, int nto,
                            uint64_t* to_rows, double* poly_f, double* poly_frac)
{
  long N = c->phim;
  long double prod_ld = 1.0L;
  for (int i = 0; i < nfrom; i++)
    prod_ld *= (long double)c->mod[from_idx[i]]->q;
  int nl = limbs_for(c, from_idx, nfrom, 8);
  crt_tab t;
  crt_tab_init(&t, c, from_idx, nfrom, nl);
  uint64_t* remtab = ifft_to_remtab(c, from_idx, nfrom, from_rows); /* toPoly */
  uint64_t* v = (uint64_t*)malloc((size_t)nl * 8);
  uint64_t* tmp = (uint64_t*)malloc((size_t)nl * 8);
  uint64_t* coef = (uint64_t*)malloc((size_t)N * nto * 8);
  for (long h = 0; h < N; h++) {
    crt_one(&t, remtab + (size_t)h * nfrom, 0, v);
    if (poly_f)
      poly_f[h] = bn_to_double_signed(v, nl, tmp);
    if (poly_frac)
      poly_frac[h] = (double)(bn_to_ldouble_signed(v, nl, tmp) / prod_ld);
    /* FFT(poly, s1): "convert(tmp, x)" reduces each bignum mod q
     * (src/CModulus.cpp:446-460) */
    for (int r = 0; r < nto; r++)
      coef[(size_t)r * N + h] = bn_smod_word(v, nl, c->mod[to_idx[r]]->q, tmp);
  }
  /* NOTE: the reference skips the FFT for constant polynomials
   * (src/DoubleCRT.cpp:595-598); the transform of a constant is the
   * constant in every slot, so the result is identical. */
  ho_dcrt_fft(c, to_idx, nto, coef, to_rows);
  free(coef);
  free(tmp);
  free(v);
  free(remtab);
  crt_tab_free(&t);
}

void ho_dcrt_scale_by_primes(const ho_ctx* c, const int* from_idx, int nfrom,
                             uint64_t* rows, const int* add_idx, int nadd)
{
  /* src/DoubleCRT.cpp:617-636 */
  long N = c->phim;
  for (int r = 0; r < nfrom; r++) {
    uint64_t qi = c->mod[from_idx[r]]->q;
    uint64_t f = 1;
    for (int a = 0; a < nadd; a++)
      f = ho_mulmod(f, c->mod[add_idx[a]]->q % qi, qi);
    ho_row_mul_scalar(rows + (size_t)r * N, rows + (size_t)r * N, f, N, qi);
  }
}

static int find_idx(const int* arr, int n, int v)
{
  for (int i = 0; i < n; i++)
    if (arr[i] == v)
      return i;
  return -1;
}

/* ------------------------------------------------------------------ */
/* breakIntoDigits  (src/DoubleCRT.cpp:479-561)                         */
/* ------------------------------------------------------------------ */
void ho_dcrt_break_into_digits(const ho_ctx* c, const int* own_idx, int nown,
                               const uint64_t* rows, const int* dig_idx,
                               const int* dig_off, int ndig,
                               const int* all_idx, int nall, uint64_t* digits)
{
  ho_dcrt_break_into_digits_norms(c, own_idx, nown, rows, dig_idx, dig_off, ndig,
                                  all_idx, nall, digits, NULL);
}
void ho_dcrt_break_into_digits_norms(const ho_ctx* c, const int* own_idx, int nown,
                                     const uint64_t* rows, const int* dig_idx,
                                     const int* dig_off, int ndig,
                                     const int* all_idx, int nall,
                                     uint64_t* digits, double* frac_norms)
{
  long N = c->phim;
  double* frac = frac_norms ? (double*)malloc((size_t)N * sizeof(double)) : NULL;
  size_t dstride = (size_t)nall * N;
  /* :509-513  digits[i] = *this restricted to the digit's primes.  We keep
   * each digit's own rows in place inside its [nall][N] output block. */
  for (int d = 0; d < ndig; d++) {
    for (int p = dig_off[d]; p < dig_off[d + 1]; p++) {
      int src = find_idx(own_idx, nown, dig_idx[p]);
      int dst = find_idx(all_idx, nall, dig_idx[p]);
      memcpy(digits + d * dstride + (size_t)dst * N, rows + (size_t)src * N,
             (size_t)N * 8);
    }
  }
  for (int d = 0; d < ndig; d++) {
    int nd = dig_off[d + 1] - dig_off[d];
    const int* didx = dig_idx + dig_off[d];
    /* gather the digit's own rows, list the primes not in the digit */
    uint64_t* own = (uint64_t*)malloc((size_t)nd * N * 8);
    int* to = (int*)malloc((size_t)nall * sizeof(int));
    int nto = 0;
    for (int p = 0; p < nd; p++) {
      int pos = find_idx(all_idx, nall, didx[p]);
      memcpy(own + (size_t)p * N, digits + d * dstride + (size_t)pos * N,
             (size_t)N * 8);
    }
    for (int r = 0; r < nall; r++)
      if (find_idx(didx, nd, all_idx[r]) < 0)
        to[nto++] = all_idx[r];
    uint64_t* ext = (uint64_t*)malloc((size_t)nto * N * 8);
    add_primes_impl(c, didx, nd, own, to, nto, ext, NULL, frac); /* :535 */
    if (frac_norms) /* :541 norm_val = embeddingLargestCoeff(poly) */
      frac_norms[d] = ho_embedding_largest_coeff(c->m, frac, N);
    for (int r = 0; r < nto; r++) {
      int pos = find_idx(all_idx, nall, to[r]);
      memcpy(digits + d * dstride + (size_t)pos * N, ext + (size_t)r * N,
             (size_t)N * 8);
    }
    /* :551-556  digits[j] -= digits[i]; digits[j] /= pi  on digit j's own
     * primes only (matchIndexSets=false) */
    for (int j = d + 1; j < ndig; j++) {
      for (int p = dig_off[j]; p < dig_off[j + 1]; p++) {
        int pos = find_idx(all_idx, nall, dig_idx[p]);
        uint64_t qj = c->mod[dig_idx[p]]->q;
        uint64_t pi = 1;
        for (int s = 0; s < nd; s++)
          pi = ho_mulmod(pi, c->mod[didx[s]]->q % qj, qj);
        uint64_t pinv = ho_invmod(pi, qj);
        uint64_t* rj = digits + j * dstride + (size_t)pos * N;
        const uint64_t* ri = digits + d * dstride + (size_t)pos * N;
        ho_row_sub(rj, rj, ri, N, qj);
        ho_row_mul_scalar(rj, rj, pinv, N, qj);
      }
    }
    free(ext);
    free(to);
    free(own);
  }
  free(frac);
}

/* ------------------------------------------------------------------ */
/* scaleDownToSet  (src/DoubleCRT.cpp:1464-1516)                        */
/* ------------------------------------------------------------------ */
void ho_dcrt_scale_down(const ho_ctx* c, const int* own_idx, int nown,
                        const uint64_t* rows, const int* drop_idx, int ndrop,
                        uint64_t ptxt_space, uint64_t* out_rows, double* fdelta)
{
  long N = c->phim;
  int nl = limbs_for(c, drop_idx, ndrop, 72);
  crt_tab t;
  crt_tab_init(&t, c, drop_idx, ndrop, nl);
  /* toPoly(delta, diff) */
  uint64_t* drows = (uint64_t*)malloc((size_t)ndrop * N * 8);
  for (int r = 0; r < ndrop; r++) {
    int src = find_idx(own_idx, nown, drop_idx[r]);
    memcpy(drows + (size_t)r * N, rows + (size_t)src * N, (size_t)N * 8);
  }
  uint64_t* remtab = ifft_to_remtab(c, drop_idx, ndrop, drows);
  int nkeep = 0;
  int* keep = (int*)malloc((size_t)nown * sizeof(int));
  for (int r = 0; r < nown; r++)
    if (find_idx(drop_idx, ndrop, own_idx[r]) < 0)
      keep[nkeep++] = own_idx[r];

  uint64_t* v = (uint64_t*)malloc((size_t)nl * 8);
  uint64_t* tmp = (uint64_t*)malloc((size_t)nl * 8);
  uint64_t* dcoef = (uint64_t*)malloc((size_t)nkeep * N * 8);
  uint64_t prodInv = 0;
  uint64_t p_over_2 = ptxt_space / 2, p_mod_2 = ptxt_space % 2;
  if (ptxt_space > 1)
    prodInv = ho_invmod(bn_mod_word(t.prod, nl, ptxt_space), ptxt_space);
  long double prod_ld = 0;
  for (int i = nl - 1; i >= 0; i--)
    prod_ld = prod_ld * 18446744073709551616.0L + (long double)t.prod[i];

  for (long h = 0; h < N; h++) {
    crt_one(&t, remtab + (size_t)h * ndrop, 0, v);
    if (ptxt_space > 1) {
      /* :1485-1508 (NTL rem(ZZ,long) is the non-negative remainder) */
      uint64_t dm = bn_smod_word(v, nl, ptxt_space, tmp);
      if (dm != 0) {
        dm = ho_mulmod(dm, prodInv, ptxt_space);
        int neg = bn_is_neg(v, nl);
        int sub_p = 0;
        if (dm > p_over_2 || (p_mod_2 == 0 && dm == p_over_2 && neg))
          sub_p = 1;
        if (sub_p) {
          /* delta_i_modP -= ptxtSpace  => negative multiplier */
          uint64_t mult = ptxt_space - dm; /* delta += diffProd * mult */
          bn_addmul_word(v, t.prod, nl, mult);
        } else {
          /* delta -= diffProd * dm */
          bn_copy(tmp, t.prod, nl);
          bn_mul_word(tmp, nl, dm);
          bn_sub(v, tmp, nl);
        }
      }
    }
    if (fdelta)
      fdelta[h] = (double)(bn_to_ldouble_signed(v, nl, tmp) / prod_ld);   /* (xdouble in the reference: delta itself exceeds the range of a double from 18 dropped primes on) */
    for (int r = 0; r < nkeep; r++)
      dcoef[(size_t)r * N + h] = bn_smod_word(v, nl, c->mod[keep[r]]->q, tmp);
  }
  /* *this -= delta (FFT of delta on kept primes), then /= diffProd */
  uint64_t* deval = (uint64_t*)malloc((size_t)nkeep * N * 8);
  ho_dcrt_fft(c, keep, nkeep, dcoef, deval);
  for (int r = 0; r < nkeep; r++) {
    uint64_t q = c->mod[keep[r]]->q;
    int src = find_idx(own_idx, nown, keep[r]);
    uint64_t dp = bn_mod_word(t.prod, nl, q);
    uint64_t dinv = ho_invmod(dp, q);
    uint64_t* o = out_rows + (size_t)r * N;
    ho_row_sub(o, rows + (size_t)src * N, deval + (size_t)r * N, N, q);
    ho_row_mul_scalar(o, o, dinv, N, q);
  }
  free(deval);
  free(dcoef);
  free(tmp);
  free(v);
  free(keep);
  free(remtab);
  free(drows);
  crt_tab_free(&t);
}

/* ------------------------------------------------------------------ */
/* tensor / key switch / multiply                                       */
/* ------------------------------------------------------------------ */
void ho_tensor(const ho_ctx* c, const int* idx, int nrows, const uint64_t* c0,
               const uint64_t* c1, const uint64_t* d0, const uint64_t* d1,
               uint64_t* o0, uint64_t* o1, uint64_t* o2)
{
  long N = c->phim;
  uint64_t* tmp = (uint64_t*)malloc((size_t)N * 8);
  for (int r = 0; r < nrows; r++) {
    uint64_t q = c->mod[idx[r]]->q;
    size_t o = (size_t)r * N;
    ho_row_mul(o0 + o, c0 + o, d0 + o, N, q);
    ho_row_mul(o1 + o, c0 + o, d1 + o, N, q);
    ho_row_mul(tmp, c1 + o, d0 + o, N, q);
    ho_row_add(o1 + o, o1 + o, tmp, N, q);
    ho_row_mul(o2 + o, c1 + o, d1 + o, N, q);
  }
  free(tmp);
}

void ho_key_switch_digits(const ho_ctx* c, const int* all_idx, int nall,
                          int ndig, const uint64_t* digits,
                          const uint64_t* ksk_b, const uint64_t* ksk_a,
                          uint64_t* out0, uint64_t* out1)
{
  long N = c->phim;
  uint64_t* tmp = (uint64_t*)malloc((size_t)N * 8);
  for (int d = 0; d < ndig; d++) {
    for (int r = 0; r < nall; r++) {
      uint64_t q = c->mod[all_idx[r]]->q;
      size_t o = ((size_t)d * nall + r) * N;
      size_t oo = (size_t)r * N;
      ho_row_mul(tmp, digits + o, ksk_a + o, N, q); /* KS_loop_1 */
      ho_row_add(out1 + oo, out1 + oo, tmp, N, q);  /* KS_loop_2 */
      ho_row_mul(tmp, digits + o, ksk_b + o, N, q); /* KS_loop_3 */
      ho_row_add(out0 + oo, out0 + oo, tmp, N, q);  /* KS_loop_4 */
    }
  }
  free(tmp);
}

void ho_mul_relin(const ho_ctx* c, const int* own_idx, int nown,
                  const int* sp_idx, int nsp, const int* dig_idx,
                  const int* dig_off, int ndig, const uint64_t* c0,
                  const uint64_t* c1, const uint64_t* d0, const uint64_t* d1,
                  const uint64_t* ksk_b, const uint64_t* ksk_a, uint64_t* out0,
                  uint64_t* out1)
{
  long N = c->phim;
  int nall = nown + nsp;
  int* all_idx = (int*)malloc((size_t)nall * sizeof(int));
  memcpy(all_idx, own_idx, (size_t)nown * sizeof(int));
  memcpy(all_idx + nown, sp_idx, (size_t)nsp * sizeof(int));
  size_t psz = (size_t)nown * N;
  uint64_t* t0 = (uint64_t*)malloc(psz * 8);
  uint64_t* t1 = (uint64_t*)malloc(psz * 8);
  uint64_t* t2 = (uint64_t*)malloc(psz * 8);
  ho_tensor(c, own_idx, nown, c0, c1, d0, d1, t0, t1, t2);
  /* reLinearize: parts (1),(s) -> addPrimesAndScale(special) (:764-767) */
  ho_dcrt_scale_by_primes(c, own_idx, nown, t0, sp_idx, nsp);
  ho_dcrt_scale_by_primes(c, own_idx, nown, t1, sp_idx, nsp);
  memset(out0, 0, (size_t)nall * N * 8);
  memset(out1, 0, (size_t)nall * N * 8);
  memcpy(out0, t0, psz * 8);
  memcpy(out1, t1, psz * 8);
  /* part s^2 -> keySwitchPart (:805-842) */
  uint64_t* digits = (uint64_t*)malloc((size_t)ndig * nall * N * 8);
  ho_dcrt_break_into_digits(c, own_idx, nown, t2, dig_idx, dig_off, ndig,
                            all_idx, nall, digits);
  ho_key_switch_digits(c, all_idx, nall, ndig, digits, ksk_b, ksk_a, out0, out1);
  free(digits);
  free(t0);
  free(t1);
  free(t2);
  free(all_idx);
}

/* ------------------------------------------------------------------ */
static uint64_t splitmix64(uint64_t* s)
{
  uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

void ho_fill_uniform(uint64_t* out, long n, uint64_t q, uint64_t seed)
{
  uint64_t s = seed;
  int bits = 0;
  while (bits < 64 && (q - 1) >> bits)
    bits++;
  uint64_t mask = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1);
  for (long i = 0; i < n; i++) {
    uint64_t v;
    do {
      v = splitmix64(&s) & mask;
    } while (v >= q);
    out[i] = v;
  }
}

/* ------------------------------------------------------------------ */
/* embeddingLargestCoeff  (src/norms.cpp:129-262, 480-493)              */
/* ------------------------------------------------------------------ */
double ho_embedding_largest_coeff(uint64_t m, const double* f, long n)
{
  const long double two_pi = 6.283185307179586476925286766559005768394L;
  long double best = 0;
  if (m >= 2 && (m & (m - 1)) == 0) {
    /* "odd-power trick" (src/norms.cpp:159-198): g_i = f_i W^i, then an (m/2)-point DFT with
     * root W^2 gives f(W^(2j+1)) for every j; all odd powers are in Z_m^*. */
    long N = (long)(m / 2);
    if (N < 1)
      N = 1;
    long double* re = (long double*)calloc((size_t)N, sizeof(long double));
    long double* im = (long double*)calloc((size_t)N, sizeof(long double));
    for (long i = 0; i < n && i < N; i++) {
      long double ang = two_pi * (long double)i / (long double)m;
      re[i] = (long double)f[i] * cosl(ang);
      im[i] = (long double)f[i] * sinl(ang);
    }
    /* in-place decimation-in-frequency, output order irrelevant for a maximum */
    for (long len = N / 2; len >= 1; len /= 2) {
      for (long blk = 0; blk < N; blk += 2 * len) {
        for (long j = 0; j < len; j++) {
          long double ang = two_pi * (long double)j / (long double)(2 * len);
          long double wr = cosl(ang), wi = sinl(ang);
          long a = blk + j, b = a + len;
          long double dr = re[a] - re[b], di = im[a] - im[b];
          re[a] += re[b];
          im[a] += im[b];
          re[b] = dr * wr - di * wi;
          im[b] = dr * wi + di * wr;
        }
      }
    }
    for (long i = 0; i < N; i++) {
      long double v = re[i] * re[i] + im[i] * im[i];
      if (v > best)
        best = v;
    }
    free(re);
    free(im);
    return (double)sqrtl(best);
  }
  /* general m: the definition (src/norms.cpp:129-157), j in Z_m^*, j <= m/2 */
  long double* cw = (long double*)malloc((size_t)m * sizeof(long double));
  long double* sw = (long double*)malloc((size_t)m * sizeof(long double));
  for (uint64_t k = 0; k < m; k++) {
    long double ang = two_pi * (long double)k / (long double)m;
    cw[k] = cosl(ang);
    sw[k] = sinl(ang);
  }
  for (uint64_t j = 1; j <= m / 2 || j == 1; j++) {
    if (gcd_u64(j, m) != 1)
      continue;
    long double ar = 0, ai = 0;
    uint64_t e = 0;
    for (long i = 0; i < n; i++) {
      ar += (long double)f[i] * cw[e];
      ai += (long double)f[i] * sw[e];
      e += j;
      if (e >= m)
        e -= m;
    }
    long double v = ar * ar + ai * ai;
    if (v > best)
      best = v;
  }
  free(cw);
  free(sw);
  return (double)sqrtl(best);
}

/* ------------------------------------------------------------------ */
/* DoubleCRT::randomize (src/DoubleCRT.cpp:1258-1378): every row is     */
/* filled by rejection sampling from a byte stream taken in 2048-byte   */
/* buffers: nb = ceil(k/8) bytes per candidate (little endian), masked  */
/* to k = NumBits(q-1) bits, kept when < q, until phi(m) values are     */
/* there; the rest of the last buffer is discarded.  The reference's    */
/* stream is NTL's RandomStream (ChaCha20 keyed by NTL's own seed       */
/* expansion, unreproducible without NTL).  Here the stream of a row is */
/* the RFC 8439 ChaCha20 key stream under the caller's 256-bit key with */
/* nonce (stream_lo, stream_hi, prime index | batch element << 16) and  */
/* block counter 0.. -- one independent stream per row, which is what   */
/* lets the device fill all rows at once.  Same acceptance rule, same   */
/* byte order, same buffer discipline as the reference.                 */
/* ------------------------------------------------------------------ */
#define HO_ROTL32(x, n) (((x) << (n)) | ((x) >> (32 - (n))))
#define HO_QR(a, b, c, d)        \
  do {                           \
    a += b; d ^= a; d = HO_ROTL32(d, 16); \
    c += d; b ^= c; b = HO_ROTL32(b, 12); \
    a += b; d ^= a; d = HO_ROTL32(d, 8);  \
    c += d; b ^= c; b = HO_ROTL32(b, 7);  \
  } while (0)

/* RFC 8439 section 2.3: key = 8 LE words, counter, nonce = 3 LE words -> 64 bytes */
void ho_chacha20_block(const uint32_t key[8], uint32_t counter, const uint32_t nonce[3], uint8_t out[64])
{
  uint32_t s[16], x[16];
  s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
  for (int i = 0; i < 8; i++)
    s[4 + i] = key[i];
  s[12] = counter;
  s[13] = nonce[0]; s[14] = nonce[1]; s[15] = nonce[2];
  memcpy(x, s, sizeof x);
  for (int r = 0; r < 10; r++) {
    HO_QR(x[0], x[4], x[8], x[12]);
    HO_QR(x[1], x[5], x[9], x[13]);
    HO_QR(x[2], x[6], x[10], x[14]);
    HO_QR(x[3], x[7], x[11], x[15]);
    HO_QR(x[0], x[5], x[10], x[15]);
    HO_QR(x[1], x[6], x[11], x[12]);
    HO_QR(x[2], x[7], x[8], x[13]);
    HO_QR(x[3], x[4], x[9], x[14]);
  }
  for (int i = 0; i < 16; i++) {
    uint32_t v = x[i] + s[i];
    out[4 * i] = (uint8_t)v;
    out[4 * i + 1] = (uint8_t)(v >> 8);
    out[4 * i + 2] = (uint8_t)(v >> 16);
    out[4 * i + 3] = (uint8_t)(v >> 24);
  }
}

/* one row: returns the number of 2048-byte buffers consumed */
long ho_randomize_row(uint64_t* row, long phim, uint64_t q, const uint32_t key[8], const uint32_t nonce[3])
{
  enum { BUFSZ = 2048 };
  uint8_t buf[BUFSZ];
  int k = 0;
  for (uint64_t t = q - 1; t; t >>= 1)
    k++;                                   /* NTL::NumBits(pi - 1) */
  const long nb = (k + 7) / 8;
  const uint64_t mask = k >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << k) - 1);
  long j = 0, nbuf = 0;
  uint32_t ctr = 0;
  if (phim <= 0)
    return 0;
  for (;;) {
    for (int b = 0; b < BUFSZ / 64; b++)
      ho_chacha20_block(key, ctr++, nonce, buf + 64 * b);   /* stream.get(buf, bufsz) */
    nbuf++;
    for (long pos = 0; pos <= BUFSZ - nb; pos += nb) {
      uint64_t u = 0;
      for (long c = nb - 1; c >= 0; c--)
        u = (u << 8) | buf[pos + c];
      u &= mask;
      row[j] = u;
      j += (u < q);
      if (j >= phim)
        return nbuf;
    }
  }
}

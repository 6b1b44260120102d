// pfa_core.h -- Cmodulus::FFT / iFFT for m = 21845 = 5 * 17 * 257 (BASELINE config 5) WITHOUT Bluestein:
// Good-Thomas x Rader.  The reference has only Bluestein for a general m (src/bluestein.cpp:134-201,
// src/CModulus.cpp:431-443, 555-577); parity is on the VALUES  y[rank(j)] = f(zeta^j), j in Z_m^* increasing,
// zeta = root^2 (the same root the Bluestein path takes), and  X[i] = sum_j y_j zeta^(-ij), i < m,  for the inverse
// (rem Phi_m and the 1/m stay where they were: conv_kernels.hip).
//
// The three prime factors of m are Fermat primes, so
//   * Good-Thomas: i = i1 M1 + i2 M2 + i3 M3 mod m (M_k = m / p_k) on the coefficient side, j = CRT(j1, j2, j3) on the
//     evaluation side turn the length-m transform into DFT_5 (x) DFT_17 (x) DFT_257 with no twiddles in between;
//   * j in Z_m^*  <=>  every j_k != 0: exactly the outputs Rader's convolution produces, and that convolution has
//     length p - 1 = 4, 16, 256 -- powers of two, no padding (the chain primes are c 2^36 + 1: the roots exist).
// A row costs ~0.19 M modular multiplications against ~1.2 M for the 2^16-point chirp convolution, its working set
// is 140 KB of LDS against a 512 KiB global intermediate, and it is ONE launch.
// tests/pfa_ref.py is the python-integer restatement this file follows index for index; tests/cpp/pfa_replay.cpp runs
// the phase functions below thread by thread on the CPU (with every compile-time bound asserted) against the oracle.
//
// One workgroup of 1024 threads = one (row, batch element).  Phases are separated by workgroup barriers; a thread's
// registers (St) carry over.  LDS (64-bit words) holds, one after the other in the same 140 KB:
//   X  [16384]            the row as loaded (coefficients forward, evaluations inverse)
//   T  [4][17][257]       after the 5-point dimension:   T[j1-1][i2][i3]
//   R  [64][ROW = 272]    after the 17-point dimension:  row c = (j1-1) 16 + b2, the 256-point convolutions run in place;
//                         an element idx sits at idx + (idx >> 4) while the convolution runs (both the stride-1 and the
//                         stride-16 access of its 16 threads are then conflict-free), at its natural i3 otherwise
//   Z0 [64]               forward: the i3 = 0 plane after the 17-point dimension (Rader's s[0] of the last dimension)
// Arithmetic: the Proth-form Montgomery product of ntt_core.h (every table word is w 2^64 mod q), values lazy in
// [0, B q) with B tracked at compile time (dif_b / dit_b below are the schedules, HX_BOUND checks them in the replay).
#pragma once
#include "ntt_core.h"

namespace hx {
namespace pfa {

constexpr int P1 = 5, P2 = 17, P3 = 257;
constexpr int M = P1 * P2 * P3;          // 21845
constexpr int PHI = 4 * 16 * 256;        // 16384
constexpr int M1 = M / P1, M2 = M / P2, M3 = M / P3;   // 4369, 1285, 85
constexpr int GEN = 3;                   // generates Z_5^*, Z_17^*, Z_257^*
constexpr int NT = 1024;
constexpr int ROW = 272;                 // = 16 mod 32: two rows' accesses in one half-wave fall into disjoint banks
constexpr int T_WORDS = 4 * P2 * P3;     // 17476
constexpr int Z0_OFF = T_WORDS;          // (R needs 64 * 272 = 17408 <= 17476)
constexpr int LDS_WORDS = Z0_OFF + 64;   // 17540 words = 140320 B  (forward)
constexpr int OUT_LDS = 20480;           // inverse: all 160 KiB stage the output row (21845 words: the last 1365 go straight to memory)
constexpr int INV_LDS_WORDS = OUT_LDS;
constexpr int NUNITS = P2 * P3;          // 4369 five-point units (i2, i3)
constexpr int UROUNDS = (NUNITS + NT - 1) / NT;   // 5

constexpr int cpowmod(int b, int e, int p)
{
  int r = 1;
  for (int i = 0; i < e; i++)
    r = r * b % p;
  return r;
}
constexpr int cinv(int a, int p) { return cpowmod(a, p - 2, p); }
// i = g^a and j = g^-b in the two small dimensions (compile-time permutations of registers)
constexpr int gpow1(int a) { return cpowmod(GEN, a, P1); }
constexpr int gipow1(int b) { return cpowmod(cinv(GEN, P1), b, P1); }
constexpr int gpow2(int a) { return cpowmod(GEN, a, P2); }
constexpr int gipow2(int b) { return cpowmod(cinv(GEN, P2), b, P2); }

// per-prime table: offsets in 64-bit words, every entry w 2^64 mod q
enum : int {
  TW4 = 0,             // [4]    rho_4^e
  TW16 = 4,            // [16]   rho_16^e
  TW256 = 20,          // [256]  rho_256^e           (rho_16 = rho_256^16, rho_4 = rho_256^64)
  F_V1 = 276,          // [4]    forward: bit-reversed spectrum of v_c = omega_1^(g^-c), over 4
  F_V2 = 280,          // [16]
  F_V3 = 296,          // [256]
  F_W2 = 552,          // [16][16] omega_2^(i2 g^-b2) at [b2][i2 - 1]   (the i3 = 0 plane, direct)
  I_V1 = 808,          // inverse: spectra of v'_c = omega^(-g^c)
  I_V2 = 812,
  I_V3 = 828,
  I_W2 = 1084,         // [17][16] omega_2^(-i2 g^-b2) at [i2][b2]      (i2 = 0: ones)
  I_MINV = 1084 + 17 * 16,   // m^-1                                    (src/CModulus.cpp:574-577)
  TAB_WORDS = I_MINV + 4
};

struct Args {
  const uint64_t* tab;     // this row's prime table
  const uint16_t* pos2;    // [64][16][16]: rank of CRT(j1, g^-b2, g^-b3) in Z_m^*, at [(c 16 + t) 16 + k], b3 = t + 16 k
  const uint16_t* dlog3;   // [257]: a with g^a = i3
  const uint16_t* gpow3;   // [256]: g^a mod 257
  const uint64_t* src;     // the row (forward: coefficients; inverse: evaluations), 16384 words
  uint64_t* dst;           // forward: the row; inverse: X, m words (stride mpad between rows)
};
struct St {
  uint64_t e[20];
  uint64_t aux;
  uint64_t o[25];   // inverse, fused rem Phi_m: the thread's 25 words of X (five 5-point units), kept to the end
  uint64_t v[6];    // ... and the values a thread carries from the read half of a multiply pass to its write half
};

// ---- the modular product, by the form of the prime ----
// QCP: a Proth-form prime (q = 1 mod 2^32: every 60-bit PrimeGenerator prime of this ring) -- the word-wise Montgomery
// product of ntt_core.h, six multiply-adds.  QCG: any other odd prime (the 40-bit small primes of a chain are
// t m 2^k + 1 with k < 32) -- the textbook Montgomery product with q' = -q^-1 mod 2^64: T = y W, (T + (T q' mod 2^64) q) / 2^64,
// in (0, 2q) for ANY 64-bit y and W < q, so every bound of the schedules below holds for it as well and the tables
// (w 2^64 mod q) are the same.  The choice is per row (workgroup-uniform), made once by the kernel.
struct QCP : QC {};
struct QCG : QC {
  uint64_t qinv;   // -q^-1 mod 2^64
};
HXD uint64_t neg_qinv64(uint64_t q)
{
  uint64_t x = q;                 // q x = 1 mod 2^3 for odd q; each step doubles the correct bits
  for (int i = 0; i < 5; i++)
    x *= 2 - q * x;
  return 0 - x;
}
HXD QCP make_qcp(uint64_t q, uint64_t mu64)
{
  QCP c;
  static_cast<QC&>(c) = make_qc(q, mu64);
  return c;
}
HXD QCG make_qcg(uint64_t q, uint64_t mu64)
{
  QCG c;
  static_cast<QC&>(c) = make_qc(q, mu64);
  c.qinv = neg_qinv64(q);
  return c;
}
HXD uint64_t pmul(uint64_t y, TWM W, const QCP& c) { return mont_mul(y, W, c); }
HXD uint64_t pacc(uint64_t y, TWM W, const QCP& c, uint64_t x) { return mont_acc(y, W, c, x); }
HXD uint64_t pmul(uint64_t y, TWM W, const QCG& c)
{
  const unsigned __int128 T = (unsigned __int128)y * W;
  const uint64_t mq = (uint64_t)T * c.qinv;
  return (uint64_t)((T + (unsigned __int128)mq * c.q) >> 64);   // T + mq q < 2^65 q: no wrap
}
HXD uint64_t pacc(uint64_t y, TWM W, const QCG& c, uint64_t x) { return x + pmul(y, W, c); }

constexpr int cpow2(int b) { return b <= 1 ? 1 : (b <= 2 ? 2 : (b <= 4 ? 4 : 8)); }
template <int K>
HXD uint64_t kq(const QC& c)
{
  static_assert(K == 1 || K == 2 || K == 4 || K == 8, "multiple of q");
  return K == 1 ? c.q : (K == 2 ? c.q2 : (K == 4 ? c.q4 : c.q8));
}
// x < B q  ->  x < T q   (T = 1, 2, 4)
template <int B, int T>
HXD uint64_t reduce(uint64_t x, const QC& c)
{
  static_assert(B >= 1 && B <= 16 && (T == 1 || T == 2 || T == 4), "bound");
  HX_BOUND(x, B, c.q);
  if constexpr (B > 8)
    x = csub(x, c.q8);
  if constexpr (B > 4 && T <= 4)
    x = csub(x, c.q4);
  if constexpr (B > 2 && T <= 2)
    x = csub(x, c.q2);
  if constexpr (B > 1 && T <= 1)
    x = csub(x, c.q);
  return x;
}

// ---- decimation in frequency (Gentleman-Sande), in place, natural order in, bit-reversed out ----
// bound of element idx after `done` stages of a 2^LOG-point pass whose inputs are all below bin q.  TRIV: the
// butterflies with twiddle exponent 0 skip the product (uniform twiddles only)
template <int LOG, bool TRIV>
constexpr int dif_b(int bin, int done, int idx)
{
  if (done == 0)
    return bin;
  const int s = done - 1, half = (1 << (LOG - 1)) >> s;
  const int pos = idx & (2 * half - 1), kk = pos & (half - 1);
  const bool upper = pos >= half;
  const int lo = upper ? idx - half : idx;
  const int bx = dif_b<LOG, TRIV>(bin, s, lo), by = dif_b<LOG, TRIV>(bin, s, lo + half);
  if (!upper)
    return bx + by > 4 ? 4 : bx + by;
  if (TRIV && kk == 0)
    return bx + cpow2(by) > 4 ? 4 : bx + cpow2(by);
  return 2;
}
template <int BX, int BY, bool MUL, class Q>
HXD void dif_bf(uint64_t& X, uint64_t& Y, TWM W, const Q& c)
{
  const uint64_t x = X, y = Y;
  HX_BOUND(x, BX, c.q);
  HX_BOUND(y, BY, c.q);
  uint64_t s = x + y;
  if constexpr (BX + BY > 4) {
    static_assert(BX + BY <= 8, "sum");
    s = csub(s, c.q4);
  }
  uint64_t d = x + kq<cpow2(BY)>(c) - y;
  constexpr int BD = BX + cpow2(BY);
  if constexpr (MUL) {
    static_assert(BD <= 12, "multiplied operand");
    d = pmul(d, W, c);
  } else if constexpr (BD > 4) {
    static_assert(BD <= 8, "difference");
    d = csub(d, c.q4);
  }
  X = s;
  Y = d;
}
// tw(IC<s>, IC<kk>) -> the twiddle of stage s, position kk within the half block
template <int LOG, int BIN, bool TRIV, class TwF, class Q>
HXD void dif_pass(uint64_t* e, TwF&& tw, const Q& c)
{
  static_for<0, LOG>([&](auto S) {
    constexpr int s = decltype(S)::value, half = (1 << (LOG - 1)) >> s;
    static_for<0, (1 << (LOG - 1))>([&](auto J) {
      constexpr int j = decltype(J)::value, kk = j % half, lo = (j / half) * 2 * half + kk, hi = lo + half;
      constexpr int bx = dif_b<LOG, TRIV>(BIN, s, lo), by = dif_b<LOG, TRIV>(BIN, s, hi);
      constexpr bool mul = !(TRIV && kk == 0);
      TWM W = 0;
      if constexpr (mul)
        W = tw(IC<s>{}, IC<kk>{});
      dif_bf<bx, by, mul>(e[lo], e[hi], W, c);
    });
  });
}
template <int LOG, bool TRIV>
constexpr int dif_out(int bin)
{
  int b = 0;
  for (int i = 0; i < (1 << LOG); i++)
    b = dif_b<LOG, TRIV>(bin, LOG, i) > b ? dif_b<LOG, TRIV>(bin, LOG, i) : b;
  return b;
}

// ---- decimation in time (Cooley-Tukey), in place, bit-reversed in, natural out, no 1/n ----
template <int LOG, bool TRIV>
constexpr int dit_b(int bin, int done, int idx)
{
  if (done == 0)
    return bin;
  const int s = done - 1, half = 1 << s;
  const int pos = idx & (2 * half - 1), kk = pos & (half - 1);
  const bool upper = pos >= half;
  const int lo = upper ? idx - half : idx;
  const int bx = dit_b<LOG, TRIV>(bin, s, lo), by = dit_b<LOG, TRIV>(bin, s, lo + half);
  if (TRIV && kk == 0) {
    const int x4 = bx > 4 ? 4 : bx, y4 = by > 4 ? 4 : by;
    return upper ? x4 + cpow2(y4) : x4 + y4;
  }
  return (bx > 6 ? 4 : bx) + 2;
}
template <int BX, int BY, bool MUL, class Q>
HXD void dit_bf(uint64_t& X, uint64_t& Y, TWM W, const Q& c)
{
  uint64_t x = X, y = Y;
  HX_BOUND(x, BX, c.q);
  HX_BOUND(y, BY, c.q);
  if constexpr (MUL) {
    static_assert(BX <= 8 && BY <= 12, "bounds");
    if constexpr (BX > 6)
      x = csub(x, c.q4);
    const uint64_t xn = pacc(y, W, c, x);   // x + R, 0 < R < 2q
    X = xn;
    Y = (x << 1) + c.q2 - xn;                   // x + 2q - R
  } else {
    static_assert(BX <= 8 && BY <= 8, "bounds");
    if constexpr (BX > 4)
      x = csub(x, c.q4);
    if constexpr (BY > 4)
      y = csub(y, c.q4);
    constexpr int Y4 = BY > 4 ? 4 : BY;
    X = x + y;
    Y = x + kq<cpow2(Y4)>(c) - y;
  }
}
template <int LOG, int BIN, bool TRIV, class TwF, class Q>
HXD void dit_pass(uint64_t* e, TwF&& tw, const Q& c)
{
  static_for<0, LOG>([&](auto S) {
    constexpr int s = decltype(S)::value, half = 1 << s;
    static_for<0, (1 << (LOG - 1))>([&](auto J) {
      constexpr int j = decltype(J)::value, kk = j % half, lo = (j / half) * 2 * half + kk, hi = lo + half;
      constexpr int bx = dit_b<LOG, TRIV>(BIN, s, lo), by = dit_b<LOG, TRIV>(BIN, s, hi);
      constexpr bool mul = !(TRIV && kk == 0);
      TWM W = 0;
      if constexpr (mul)
        W = tw(IC<s>{}, IC<kk>{});
      dit_bf<bx, by, mul>(e[lo], e[hi], W, c);
    });
  });
}
template <int LOG, bool TRIV>
constexpr int dit_out(int bin)
{
  int b = 0;
  for (int i = 0; i < (1 << LOG); i++)
    b = dit_b<LOG, TRIV>(bin, LOG, i) > b ? dit_b<LOG, TRIV>(bin, LOG, i) : b;
  return b;
}

// a cyclic convolution of length 2^LOG held by one thread: e <- IDFT(DFT(e) . hat), uniform tables.
// Returns nothing; after the forward half e[0] is the plain sum of the inputs (dc, bound DCB) for the caller that
// wants it (the inverse direction's output 0).  In: all below BIN q.  Out: element i below dit_b<LOG, true>(2, LOG, i) q.
template <int LOG, int BIN>
struct UConv {
  static constexpr int N = 1 << LOG;
  static constexpr int MID = dif_out<LOG, true>(BIN);
  static constexpr int DCB = dif_b<LOG, true>(BIN, LOG, 0);
  static constexpr int out_b(int i) { return dit_b<LOG, true>(2, LOG, i); }
  // tw: [N] powers of rho_N; hat: [N]
  template <class Q>
  static HXD void run(uint64_t* e, const TWM* tw, const TWM* hat, const Q& c, uint64_t* dc = nullptr)
  {
    dif_pass<LOG, BIN, true>(e, [&](auto S, auto K) { return tw[decltype(K)::value << decltype(S)::value]; }, c);
    if (dc)
      *dc = e[0];
    static_assert(MID <= 12, "pointwise operand");
    static_for<0, N>([&](auto I) {
      constexpr int i = decltype(I)::value;
      e[i] = pmul(e[i], hat[i], c);
    });
    dit_pass<LOG, 2, true>(
        e,
        [&](auto S, auto K) {
          constexpr int ex = decltype(K)::value * ((N / 2) >> decltype(S)::value);
          return tw[(N - ex) & (N - 1)];
        },
        c);
  }
};

// position of element idx of a 256-point convolution inside its LDS row while the convolution runs
HXD unsigned pad16(unsigned idx) { return idx + (idx >> 4); }

// the 4 lane-dependent stages of the 256-point transform on e[k] = element t + 16 k (stages of distance 128 .. 16)
template <int BIN, class Q>
HXD void dif256_outer(uint64_t* e, unsigned t, const TWM* tw256, const Q& c)
{
  dif_pass<4, BIN, false>(
      e, [&](auto S, auto K) { return tw256[(t << decltype(S)::value) + (decltype(K)::value << (4 + decltype(S)::value))]; }, c);
}
template <int BIN, class Q>
HXD void dit256_outer(uint64_t* e, unsigned t, const TWM* tw256, const Q& c)
{
  dit_pass<4, BIN, false>(
      e,
      [&](auto S, auto K) {
        constexpr int s = decltype(S)::value;
        const unsigned ex = (t + 16u * decltype(K)::value) << (3 - s);   // (t + 16 kk) * (8 / half)
        return tw256[(256u - ex) & 255u];
      },
      c);
}
constexpr int OUTER_DIF_OUT(int bin) { return dif_out<4, false>(bin); }
constexpr int OUTER_DIT_OUT(int bin) { return dit_out<4, false>(bin); }

// index arithmetic of a five-point unit u = (i2, i3): the five coefficient indices i1 M1 + i2 M2 + i3 M3 mod m
HXD unsigned unit_base(unsigned u)
{
  const unsigned i2 = u / (unsigned)P3, i3 = u - i2 * (unsigned)P3;
  return (i2 * (unsigned)M2 + i3 * (unsigned)M3) % (unsigned)M;
}
HXD unsigned wrap_m(unsigned i) { return i >= (unsigned)M ? i - (unsigned)M : i; }

// ---- the 256-point convolution of row c, shared by both directions: three phases around two in-row exchanges ----
// phase a: e[k] = element t + 16 k (already loaded, below BIN q): outer stages, park in the row
template <int BIN, class Q>
HXD void conv256_a(uint64_t* e, unsigned c, unsigned t, uint64_t* lds, const TWM* tab, const Q& q)
{
  dif256_outer<BIN>(e, t, tab + TW256, q);
  static_for<0, 16>([&](auto K) {
    constexpr int k = decltype(K)::value;
    lds[c * ROW + t + 17u * k] = e[k];
  });
}
// phase b: thread t takes elements 16 t + k: inner 16-point stages, product with the fixed spectrum, inner stages back
template <int BIN, class Q>   // BIN = OUTER_DIF_OUT(...)
HXD void conv256_b(uint64_t* e, unsigned c, unsigned t, uint64_t* lds, const TWM* tab, const TWM* hat, const Q& q, uint64_t* dc)
{
  static_for<0, 16>([&](auto K) {
    constexpr int k = decltype(K)::value;
    e[k] = lds[c * ROW + 17u * t + k];
  });
  const TWM* h = hat + 16u * t;
  const TWM* tw = tab + TW16;
  using U = UConv<4, BIN>;
  dif_pass<4, BIN, true>(e, [&](auto S, auto K) { return tw[decltype(K)::value << decltype(S)::value]; }, q);
  if (dc)
    *dc = e[0];
  static_for<0, 16>([&](auto I) {
    constexpr int i = decltype(I)::value;
    e[i] = pmul(e[i], h[i], q);
  });
  dit_pass<4, 2, true>(
      e,
      [&](auto S, auto K) {
        constexpr int ex = decltype(K)::value * (8 >> decltype(S)::value);
        return tw[(16 - ex) & 15];
      },
      q);
  static_for<0, 16>([&](auto K) {
    constexpr int k = decltype(K)::value;
    // (one common bound for the hand-over: every element below 8 q)
    HX_BOUND(e[k], U::out_b(k), q.q);
    lds[c * ROW + 17u * t + k] = e[k];
  });
}
constexpr int CONV_B_OUT = 8;
static_assert(dit_out<4, true>(2) <= CONV_B_OUT, "hand-over bound");
// phase c: back to elements t + 16 k, outer stages; element k of the result is below CONV_C_OUT q
constexpr int CONV_C_OUT = OUTER_DIT_OUT(CONV_B_OUT);
template <class Q>
HXD void conv256_c(uint64_t* e, unsigned c, unsigned t, const uint64_t* lds, const TWM* tab, const Q& q)
{
  static_for<0, 16>([&](auto K) {
    constexpr int k = decltype(K)::value;
    e[k] = lds[c * ROW + t + 17u * k];
  });
  dit256_outer<CONV_B_OUT>(e, t, tab + TW256, q);
}

// ======================================================================================================
// forward:  coefficients -> evaluations on Z_m^*
//   phase 0  X <- the row
//   phase 1  five-point units (i2, i3): Rader-5 from X, results in registers
//   phase 2  T <- them
//   phase 3  thread (i3 = 1 + (tid & 255), j1 = 1 + (tid >> 8)): Rader-17 over i2 from T, results in registers;
//            threads < 64: Z0[c] = the i3 = 0 plane's 17-point outputs, direct (16 products)
//   phase 4  R <- them (row c = (j1-1) 16 + b2, element a3 = dlog_g(i3) at its padded place)
//   phase 5..7  conv256 a, b, c; + Z0[c]; canonical
//   phase 8  X[rank] <- results (scatter inside LDS)
//   phase 9  the row <- X
// ======================================================================================================
constexpr int FWD_PHASES = 10;
constexpr int F_OUTER_IN = 2;
template <int PH, class Q>
HXD void fwd(unsigned tid, St& s, uint64_t* lds, const Args& A, const Q& q)
{
  const TWM* tab = A.tab;
  if constexpr (PH == 0) {
    static_for<0, 8>([&](auto K) {
      constexpr unsigned k = decltype(K)::value;
      const unsigned i = 2u * (k * NT + tid);
      lds[i] = A.src[i];
      lds[i + 1] = A.src[i + 1];
    });
  } else if constexpr (PH == 1) {
    static_for<0, UROUNDS>([&](auto Rr) {
      constexpr unsigned r = decltype(Rr)::value;
      const unsigned u = tid + r * NT;
      if (u < (unsigned)NUNITS) {
        const unsigned base = unit_base(u);
        uint64_t x[5];
        static_for<0, 5>([&](auto I1) {
          constexpr unsigned i1 = decltype(I1)::value;
          const unsigned i = wrap_m(base + i1 * (unsigned)M1);
          x[i1] = i < (unsigned)PHI ? lds[i] : 0;
        });
        uint64_t e[4];
        static_for<0, 4>([&](auto Aa) {
          constexpr int a = decltype(Aa)::value;
          e[a] = x[gpow1(a)];
        });
        using U = UConv<2, 1>;
        U::run(e, tab + TW4, tab + F_V1, q);
        static_for<0, 4>([&](auto B) {
          constexpr int b = decltype(B)::value;
          // output j1 = g^-b: slot j1 - 1 of this round
          s.e[r * 4 + (gipow1(b) - 1)] = reduce<U::out_b(b) + 1, 2>(x[0] + e[b], q);
        });
      }
    });
  } else if constexpr (PH == 2) {
    static_for<0, UROUNDS>([&](auto Rr) {
      constexpr unsigned r = decltype(Rr)::value;
      const unsigned u = tid + r * NT;
      if (u < (unsigned)NUNITS) {
        static_for<0, 4>([&](auto J) {
          constexpr unsigned j = decltype(J)::value;
          lds[j * (unsigned)NUNITS + u] = s.e[r * 4 + j];   // T[j1-1][i2][i3], u = i2 257 + i3
        });
      }
    });
  } else if constexpr (PH == 3) {
    const unsigned i3 = 1u + (tid & 255u), jj = tid >> 8;
    uint64_t t[17];
    static_for<0, 17>([&](auto I2) {
      constexpr unsigned i2 = decltype(I2)::value;
      t[i2] = lds[(jj * 17u + i2) * (unsigned)P3 + i3];
    });
    uint64_t z0 = 0;
    if (tid < 64u) {
      // the i3 = 0 plane: z0[c] = T[j1-1][0][0] + sum_{i2 = 1..16} T[j1-1][i2][0] omega_2^(i2 g^-b2), c = tid
      const unsigned j0 = tid >> 4, b2 = tid & 15u;
      const TWM* w2 = tab + F_W2 + 16u * b2;
      uint64_t acc = lds[(j0 * 17u) * (unsigned)P3];
      static_for<0, 16>([&](auto I) {
        constexpr int i = decltype(I)::value;
        acc = pacc(lds[(j0 * 17u + 1u + i) * (unsigned)P3], w2[i], q, acc);   // + (0, 2q)
        if constexpr (i >= 3)
          acc = csub(acc, q.q8);   // 2 + 2 (i + 1) <= 8 up to i = 2; from then on 10 -> 8
      });
      z0 = reduce<8, 2>(acc, q);
    }
    uint64_t* e = s.e;
    static_for<0, 16>([&](auto Aa) {
      constexpr int a = decltype(Aa)::value;
      e[a] = t[gpow2(a)];
    });
    using U = UConv<4, 2>;
    U::run(e, tab + TW16, tab + F_V2, q);
    static_for<0, 16>([&](auto B) {
      constexpr int b = decltype(B)::value;
      e[b] = reduce<U::out_b(b) + 2, 2>(t[0] + e[b], q);   // z[j1][b2 = b][i3]
    });
    s.aux = z0;
  } else if constexpr (PH == 4) {
    const unsigned i3 = 1u + (tid & 255u), jj = tid >> 8;
    const unsigned a3 = pad16(A.dlog3[i3]);
    static_for<0, 16>([&](auto B) {
      constexpr unsigned b = decltype(B)::value;
      lds[(jj * 16u + b) * ROW + a3] = s.e[b];
    });
    if (tid < 64u)
      lds[Z0_OFF + tid] = s.aux;
  } else if constexpr (PH == 5) {
    const unsigned c = tid >> 4, t = tid & 15u;
    static_for<0, 16>([&](auto K) {
      constexpr int k = decltype(K)::value;
      s.e[k] = lds[c * ROW + t + 17u * k];
    });
    conv256_a<F_OUTER_IN>(s.e, c, t, lds, tab, q);
  } else if constexpr (PH == 6) {
    const unsigned c = tid >> 4, t = tid & 15u;
    conv256_b<OUTER_DIF_OUT(F_OUTER_IN)>(s.e, c, t, lds, tab, tab + F_V3, q, nullptr);
  } else if constexpr (PH == 7) {
    const unsigned c = tid >> 4, t = tid & 15u;
    conv256_c(s.e, c, t, lds, tab, q);
    const uint64_t z0 = lds[Z0_OFF + c];
    static_for<0, 16>([&](auto K) {
      constexpr int k = decltype(K)::value;
      s.e[k] = reduce<CONV_C_OUT + 2, 1>(s.e[k] + z0, q);
    });
  } else if constexpr (PH == 8) {
    // 16 ranks of this thread: 32 contiguous bytes
    const uint16_t* p = A.pos2 + 16u * tid;
    static_for<0, 16>([&](auto K) {
      constexpr int k = decltype(K)::value;
      lds[p[k]] = s.e[k];
    });
  } else if constexpr (PH == 9) {
    static_for<0, 8>([&](auto K) {
      constexpr unsigned k = decltype(K)::value;
      const unsigned i = 2u * (k * NT + tid);
      A.dst[i] = lds[i];
      A.dst[i + 1] = lds[i + 1];
    });
  }
}

// ======================================================================================================
// inverse:  evaluations on Z_m^* -> X[i], i < m  (the length-m inverse transform; rem Phi_m and 1/m follow)
//   phase 0  X <- the row
//   phase 1  thread (c, t) gathers u_b3 = y[rank(CRT(j1, g^-b2, g^-b3))], b3 = t + 16 k
//   phase 2..4  conv256 a, b, c (b keeps the plain sum: the output i3 = 0)
//   phase 5  R <- results at their natural i3 = g^a3 (and i3 = 0)
//   phase 6  thread (i3, j1): Rader-17 transposed over b2 from R; threads < 68: the i3 = 0 column, direct
//   phase 7  T <- them
//   phase 8  five-point units (i2, i3): their four inputs from T into registers
//   phase 9  Rader-5 transposed; X[i1 M1 + i2 M2 + i3 M3 mod m] -> LDS (the first 20480 words) or memory
//   phase 10 the row <- LDS
// ======================================================================================================
constexpr int INV_PHASES = 11;
constexpr int I_OUTER_IN = 1;
template <int PH, class Q>
HXD void inv(unsigned tid, St& s, uint64_t* lds, const Args& A, const Q& q)
{
  const TWM* tab = A.tab;
  if constexpr (PH == 0) {
    fwd<0>(tid, s, lds, A, q);
  } else if constexpr (PH == 1) {
    const uint16_t* p = A.pos2 + 16u * tid;
    static_for<0, 16>([&](auto K) {
      constexpr int k = decltype(K)::value;
      s.e[k] = lds[p[k]];
    });
  } else if constexpr (PH == 2) {
    const unsigned c = tid >> 4, t = tid & 15u;
    conv256_a<I_OUTER_IN>(s.e, c, t, lds, tab, q);
  } else if constexpr (PH == 3) {
    const unsigned c = tid >> 4, t = tid & 15u;
    uint64_t dc;
    conv256_b<OUTER_DIF_OUT(I_OUTER_IN)>(s.e, c, t, lds, tab, tab + I_V3, q, &dc);
    s.aux = reduce<dif_b<4, true>(OUTER_DIF_OUT(I_OUTER_IN), 4, 0), 2>(dc, q);   // (thread t = 0: the sum of all 256)
  } else if constexpr (PH == 4) {
    const unsigned c = tid >> 4, t = tid & 15u;
    conv256_c(s.e, c, t, lds, tab, q);
    static_for<0, 16>([&](auto K) {
      constexpr int k = decltype(K)::value;
      s.e[k] = reduce<CONV_C_OUT, 2>(s.e[k], q);
    });
  } else if constexpr (PH == 5) {
    const unsigned c = tid >> 4, t = tid & 15u;
    static_for<0, 16>([&](auto K) {
      constexpr unsigned k = decltype(K)::value;
      lds[c * ROW + A.gpow3[t + 16u * k]] = s.e[k];   // w[c][i3 = g^a3]
    });
    if (t == 0)
      lds[c * ROW] = s.aux;                           // w[c][0]
  } else if constexpr (PH == 6) {
    const unsigned i3 = 1u + (tid & 255u), jj = tid >> 8;
    uint64_t* e = s.e;
    static_for<0, 16>([&](auto B) {
      constexpr unsigned b = decltype(B)::value;
      e[b] = lds[(jj * 16u + b) * ROW + i3];
    });
    uint64_t col = 0;
    if (tid < 68u) {
      // the i3 = 0 column: t[j0][i2][0] = sum_b2 w[j0][b2][0] omega_2^(-i2 g^-b2)
      const unsigned j0 = tid / 17u, i2 = tid - j0 * 17u;
      const TWM* w2 = tab + I_W2 + 16u * i2;
      uint64_t acc = 0;
      static_for<0, 16>([&](auto B) {
        constexpr int b = decltype(B)::value;
        acc = pacc(lds[(j0 * 16u + b) * ROW], w2[b], q, acc);
        if constexpr (b >= 3)
          acc = csub(acc, q.q8);
      });
      col = reduce<8, 2>(acc, q);
    }
    using U = UConv<4, 2>;
    uint64_t dc;
    U::run(e, tab + TW16, tab + I_V2, q, &dc);
    // 17 outputs: i2 = 0 the plain sum, i2 = g^a the convolution's element a
    uint64_t t[17];
    t[0] = reduce<U::DCB, 2>(dc, q);
    static_for<0, 16>([&](auto Aa) {
      constexpr int a = decltype(Aa)::value;
      t[gpow2(a)] = reduce<U::out_b(a), 2>(e[a], q);
    });
    static_for<0, 17>([&](auto I) {
      constexpr int i = decltype(I)::value;
      s.e[i] = t[i];
    });
    s.aux = col;
  } else if constexpr (PH == 7) {
    const unsigned i3 = 1u + (tid & 255u), jj = tid >> 8;
    static_for<0, 17>([&](auto I2) {
      constexpr unsigned i2 = decltype(I2)::value;
      lds[(jj * 17u + i2) * (unsigned)P3 + i3] = s.e[i2];
    });
    if (tid < 68u)
      lds[tid * (unsigned)P3] = s.aux;   // (j0 17 + i2) 257 + 0
  } else if constexpr (PH == 8) {
    // (the units' inputs leave T for the registers first: T's LDS becomes the staging area of the output)
    static_for<0, UROUNDS>([&](auto Rr) {
      constexpr unsigned r = decltype(Rr)::value;
      const unsigned u = tid + r * NT;
      if (u < (unsigned)NUNITS) {
        static_for<0, 4>([&](auto B) {
          constexpr int b = decltype(B)::value;
          s.e[r * 4 + b] = lds[(unsigned)(gipow1(b) - 1) * (unsigned)NUNITS + u];   // u'_b = T[j1 = g^-b]
        });
      }
    });
  } else if constexpr (PH == 9) {
    // X[i], i = i1 M1 + i2 M2 + i3 M3 mod m, is scattered over the whole row: through the LDS (all 160 KiB of it hold
    // the first OUT_LDS = 20480 of the m = 21845 words; the 1365 beyond go to memory as they are), so that the row
    // leaves in 16-byte coalesced stores -- 8-byte stores 680 bytes apart made this kernel twice as slow as the forward one
    static_for<0, UROUNDS>([&](auto Rr) {
      constexpr unsigned r = decltype(Rr)::value;
      const unsigned u = tid + r * NT;
      if (u < (unsigned)NUNITS) {
        uint64_t e[4];
        static_for<0, 4>([&](auto B) {
          constexpr int b = decltype(B)::value;
          e[b] = s.e[r * 4 + b];
        });
        using U = UConv<2, 2>;
        uint64_t dc;
        U::run(e, tab + TW4, tab + I_V1, q, &dc);
        const unsigned base = unit_base(u);
        auto put = [&](unsigned i, uint64_t v) {
          if (i < (unsigned)OUT_LDS)
            lds[i] = v;
          else
            A.dst[i] = v;
        };
        put(base, reduce<U::DCB, 1>(dc, q));
        static_for<0, 4>([&](auto Aa) {
          constexpr int a = decltype(Aa)::value;
          put(wrap_m(base + (unsigned)gpow1(a) * (unsigned)M1), reduce<U::out_b(a), 1>(e[a], q));
        });
      }
    });
  } else if constexpr (PH == 10) {
    static_for<0, OUT_LDS / (2 * NT)>([&](auto K) {
      constexpr unsigned k = decltype(K)::value;
      const unsigned i = 2u * (k * NT + tid);
      A.dst[i] = lds[i];
      A.dst[i + 1] = lds[i + 1];
    });
  }
}

// ======================================================================================================
// rem Phi_m WITHOUT a multiplication, fused behind the inverse transform (phases 9 .. of inv_rem below).
//
// The reference divides by Phi_m with two FFT multiplications (src/NumbTh.cpp:1741-1804 behind src/CModulus.cpp:571);
// on this engine those were two 2^14-point convolution launches, 238 us per 512 rows against 104 for the whole
// Good-Thomas x Rader transform in front of them.  But  Phi_m = prod_{d | m} (x^d - 1)^mu(m/d), for m = 5 * 17 * 257:
//     Phi_m (x) =  (x^m - 1) (x^5 - 1) (x^17 - 1) (x^257 - 1)  /  [ (x - 1) (x^85 - 1) (x^1285 - 1) (x^4369 - 1) ]
// -- a product and quotient of BINOMIALS.  Multiplying a power series by (1 - x^d) is w_i -= w_(i-d); dividing by it is
// the running sum w_i += w_(i-d) along the d chains of stride d.  With n = phi(m), dq = m - 1 - n, t = 1/x:
//   Xr(t) = sum_k X_(m-1-k) t^k   (the top dq + 1 words of X, reversed)
//   Qr    = Xr / Phi_m(t) = Xr * prod_{A} (1 - t^d) / prod_{B} (1 - t^d)   mod t^(dq+1),  A = {1, 85, 1285, 4369}, B = {5, 17, 257}
//           (Phi_m is palindromic and 1 / (t^m - 1) = -1 mod t^m),   Q_k = Qr_(dq-k)
//   W     = Q Phi_m mod x^n = Q * prod_{B} (1 - x^d) / prod_{A} (1 - x^d)  mod x^n
//   r_i   = (X_i - W_i) / m,  i < n
// -- about 0.15 M modular additions per row in place of four 2^14-point transforms, all in the workgroup's LDS, the
// same launch.  tests/pfa_ref.py: rem_by_binomials restates it in python integers; the replay checks it word for word.
//
// A running sum of stride D over LEN words by 1024 threads: D chains; a chain is cut into segments of SEG elements,
// one (chain, segment) task per thread: (A) segment total -> AUX1; (B) totals of groups of 32 segments -> AUX2 (only
// when a chain has more than 40 segments); (C) carry-in = the groups and segments before mine, running sums written.  Chains no longer than 22 elements are summed by one thread each in a single phase.
// Everything canonical modulo q.
// ======================================================================================================
constexpr int RN = PHI, RDQ = M - 1 - PHI, RL1 = RDQ + 1;   // 16384, 5460, 5461
constexpr int AUX1 = PHI, AUX2 = PHI + NT;                   // segment totals [1024], group totals [<= 64]
constexpr int REM_LDS_WORDS = AUX2 + 64;                     // 17472
HXD uint64_t addm(uint64_t a, uint64_t b, const QC& c) { return csub(a + b, c.q); }
HXD uint64_t subm(uint64_t a, uint64_t b, const QC& c) { return csub(a + c.q - b, c.q); }

// w <- w (1 - x^DA)(1 - x^DB) (DB = 0: one binomial), logical input w_k = lds[REV ? LEN_IN - 1 - k : k], k < LEN_IN;
// output words k < LEN_OUT (<= 6 per thread), then zeros up to ZERO_TO
template <int DA, int DB, int LEN_IN, int LEN_OUT, int ZERO_TO, bool REV>
struct MulPass {
  static constexpr int PER = (LEN_OUT + NT - 1) / NT;
  static_assert(PER <= 6, "registers");
  static HXD uint64_t at(const uint64_t* w, int k) { return (k >= 0 && k < LEN_IN) ? w[REV ? LEN_IN - 1 - k : k] : 0; }
  static HXD void read(unsigned tid, St& s, const uint64_t* w, const QC& q)
  {
    static_for<0, PER>([&](auto J) {
      constexpr int j = decltype(J)::value;
      const int k = (int)tid + NT * j;
      if (k < LEN_OUT) {
        uint64_t v = subm(at(w, k), at(w, k - DA), q);
        if constexpr (DB != 0)
          v = subm(addm(v, at(w, k - DA - DB), q), at(w, k - DB), q);
        s.v[j] = v;
      }
    });
  }
  static HXD void write(unsigned tid, const St& s, uint64_t* w)
  {
    static_for<0, PER>([&](auto J) {
      constexpr int j = decltype(J)::value;
      const int k = (int)tid + NT * j;
      if (k < LEN_OUT)
        w[k] = s.v[j];
    });
    if constexpr (ZERO_TO > LEN_OUT) {
      static_for<0, (ZERO_TO - LEN_OUT + NT - 1) / NT>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const int k = LEN_OUT + (int)tid + NT * j;
        if (k < ZERO_TO)
          w[k] = 0;
      });
    }
  }
};
template <int D, int LEN, int SEG>
struct ScanPass {
  static constexpr int L = (LEN + D - 1) / D;            // longest chain
  static constexpr int NSEG = (L + SEG - 1) / SEG;
  static constexpr int NTASK = D * NSEG;
  static constexpr bool SINGLE = NSEG == 1;
  static constexpr bool GROUPS = NSEG > 40;
  static constexpr int NGRP = (NSEG + 31) / 32;
  static_assert(SINGLE || NTASK <= NT, "one task per thread");
  static_assert(SEG <= 17 || SINGLE, "registers");
  static_assert(!GROUPS || NGRP * D <= 64, "AUX2");
  // chains of at most SEG elements: one thread sums a whole chain, in place
  static HXD void single(unsigned tid, uint64_t* w, const QC& q)
  {
    static_for<0, (D + NT - 1) / NT>([&](auto Rr) {
      constexpr int r = decltype(Rr)::value;
      const int c = (int)tid + NT * r;
      if (c < D) {
        uint64_t run = 0;
        static_for<0, L>([&](auto Ll) {
          constexpr int l = decltype(Ll)::value;
          const int i = c + D * l;
          if (i < LEN) {
            run = addm(run, w[i], q);
            w[i] = run;
          }
        });
      }
    });
  }
  // task = seg * D + chain (neighbouring lanes: neighbouring chains)
  static HXD void a(unsigned tid, St&, uint64_t* w, const QC& q)
  {
    if ((int)tid < NTASK) {
      const int seg = (int)tid / D, c = (int)tid - seg * D;
      uint64_t run = 0;
      static_for<0, SEG>([&](auto Ll) {
        constexpr int l = decltype(Ll)::value;
        const int i = c + D * (seg * SEG + l);
        if (i < LEN)
          run = addm(run, w[i], q);
      });
      w[AUX1 + tid] = run;
    }
  }
  static HXD void b(unsigned tid, uint64_t* w, const QC& q)
  {
    if ((int)tid < NGRP * D) {
      const int g = (int)tid / D, c = (int)tid - g * D;
      uint64_t sum = 0;
      for (int sg = 32 * g; sg < 32 * g + 32 && sg < NSEG; sg++)
        sum = addm(sum, w[AUX1 + sg * D + c], q);
      w[AUX2 + tid] = sum;
    }
  }
  static HXD void c(unsigned tid, const St&, uint64_t* w, const QC& q)
  {
    if ((int)tid < NTASK) {
      const int seg = (int)tid / D, ch = (int)tid - seg * D;
      uint64_t carry = 0;
      int first = 0;
      if constexpr (GROUPS) {
        for (int g = 0; g < seg / 32; g++)
          carry = addm(carry, w[AUX2 + g * D + ch], q);
        first = 32 * (seg / 32);
      }
      for (int sg = first; sg < seg; sg++)
        carry = addm(carry, w[AUX1 + sg * D + ch], q);
      // (the segment is read again rather than carried over in registers: 17 LDS reads against 34 registers that,
      // next to the 50 of the thread's X words, spilled)
      static_for<0, SEG>([&](auto Ll) {
        constexpr int l = decltype(Ll)::value;
        const int i = ch + D * (seg * SEG + l);
        if (i < LEN) {
          carry = addm(carry, w[i], q);
          w[i] = carry;
        }
      });
    }
  }
};
// the passes of the two steps
using RM1a = MulPass<1, 85, RL1, RL1, 0, false>;
using RM1b = MulPass<1285, 4369, RL1, RL1, 0, false>;
using RD1a = ScanPass<5, RL1, 7>;
using RD1b = ScanPass<17, RL1, 11>;
using RD1c = ScanPass<257, RL1, 22>;
using RM2a = MulPass<5, 17, RL1, RL1 + 22, RN, true>;           // reads Q_k = Qr_(dq - k); zero beyond
using RM2b = MulPass<257, 0, RL1 + 22, RL1 + 22 + 257, 0, false>;
using RD2a = ScanPass<1, RN, 17>;
using RD2b = ScanPass<85, RN, 17>;
using RD2c = ScanPass<1285, RN, 13>;
using RD2d = ScanPass<4369, RN, 4>;
static_assert(RD1c::SINGLE && RD2c::SINGLE && RD2d::SINGLE && RD1a::GROUPS && !RD1b::GROUPS && RD2a::GROUPS && !RD2b::GROUPS, "pass shapes");

// the inverse transform with rem Phi_m and 1/m behind it: phases 0..8 are inv<0..8>; dst = the poly row
constexpr int INV_REM_PHASES = 33;
template <int PH, class Q>
HXD void inv_rem(unsigned tid, St& s, uint64_t* lds, const Args& A, const Q& q)
{
  const TWM* tab = A.tab;
  if constexpr (PH <= 8) {
    inv<PH>(tid, s, lds, A, q);
  } else if constexpr (PH == 9) {
    // the five-point units: 25 words of X per thread stay in registers; the top dq + 1 go to the LDS reversed (Xr)
    static_for<0, UROUNDS>([&](auto Rr) {
      constexpr unsigned r = decltype(Rr)::value;
      const unsigned u = tid + r * NT;
      if (u < (unsigned)NUNITS) {
        uint64_t e[4];
        static_for<0, 4>([&](auto B) {
          constexpr int b = decltype(B)::value;
          e[b] = s.e[r * 4 + b];
        });
        using U = UConv<2, 2>;
        uint64_t dc;
        U::run(e, tab + TW4, tab + I_V1, q, &dc);
        const unsigned base = unit_base(u);
        s.o[r * 5] = reduce<U::DCB, 1>(dc, q);
        if (base >= (unsigned)RN)
          lds[(unsigned)(M - 1) - base] = s.o[r * 5];
        static_for<0, 4>([&](auto Aa) {
          constexpr int a = decltype(Aa)::value;
          const unsigned i = wrap_m(base + (unsigned)gpow1(a) * (unsigned)M1);
          s.o[r * 5 + 1 + a] = reduce<U::out_b(a), 1>(e[a], q);
          if (i >= (unsigned)RN)
            lds[(unsigned)(M - 1) - i] = s.o[r * 5 + 1 + a];
        });
      }
    });
  } else if constexpr (PH == 10) {
    RM1a::read(tid, s, lds, q);
  } else if constexpr (PH == 11) {
    RM1a::write(tid, s, lds);
  } else if constexpr (PH == 12) {
    RM1b::read(tid, s, lds, q);
  } else if constexpr (PH == 13) {
    RM1b::write(tid, s, lds);
  } else if constexpr (PH == 14) {
    RD1a::a(tid, s, lds, q);
  } else if constexpr (PH == 15) {
    RD1a::b(tid, lds, q);
  } else if constexpr (PH == 16) {
    RD1a::c(tid, s, lds, q);
  } else if constexpr (PH == 17) {
    RD1b::a(tid, s, lds, q);
  } else if constexpr (PH == 18) {
    RD1b::c(tid, s, lds, q);
  } else if constexpr (PH == 19) {
    RD1c::single(tid, lds, q);
  } else if constexpr (PH == 20) {
    RM2a::read(tid, s, lds, q);
  } else if constexpr (PH == 21) {
    RM2a::write(tid, s, lds);
  } else if constexpr (PH == 22) {
    RM2b::read(tid, s, lds, q);
  } else if constexpr (PH == 23) {
    RM2b::write(tid, s, lds);
  } else if constexpr (PH == 24) {
    RD2a::a(tid, s, lds, q);
  } else if constexpr (PH == 25) {
    RD2a::b(tid, lds, q);
  } else if constexpr (PH == 26) {
    RD2a::c(tid, s, lds, q);
  } else if constexpr (PH == 27) {
    RD2b::a(tid, s, lds, q);
  } else if constexpr (PH == 28) {
    RD2b::c(tid, s, lds, q);
  } else if constexpr (PH == 29) {
    RD2c::single(tid, lds, q);
  } else if constexpr (PH == 30) {
    RD2d::single(tid, lds, q);
  } else if constexpr (PH == 31) {
    // r_i = (X_i - W_i) / m in place, by the thread that holds X_i
    const TWM minv = tab[I_MINV];
    static_for<0, UROUNDS>([&](auto Rr) {
      constexpr unsigned r = decltype(Rr)::value;
      const unsigned u = tid + r * NT;
      if (u < (unsigned)NUNITS) {
        const unsigned base = unit_base(u);
        static_for<0, 5>([&](auto I) {
          constexpr int i5 = decltype(I)::value;
          const unsigned i = i5 == 0 ? base : wrap_m(base + (unsigned)gpow1(i5 == 0 ? 0 : i5 - 1) * (unsigned)M1);
          if (i < (unsigned)RN)
            lds[i] = csub(pmul(s.o[r * 5 + i5] + q.q - lds[i], minv, q), q.q);
        });
      }
    });
  } else if constexpr (PH == 32) {
    fwd<9>(tid, s, lds, A, q);   // the row <- LDS
  }
}

// ---------------------------------------------------------------------------------------------------
// host side: tables (plain C++, 128-bit arithmetic; engine.hip and the CPU replay call the same code)
// ---------------------------------------------------------------------------------------------------
namespace host {
typedef unsigned __int128 u128;
inline uint64_t mulm(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }
inline uint64_t powm(uint64_t a, uint64_t e, uint64_t q)
{
  uint64_t r = 1 % q;
  a %= q;
  while (e) {
    if (e & 1)
      r = mulm(r, a, q);
    a = mulm(a, a, q);
    e >>= 1;
  }
  return r;
}
inline uint64_t mont(uint64_t w, uint64_t q) { return (uint64_t)((((u128)w) << 64) % q); }
inline unsigned brev(unsigned x, int bits)
{
  unsigned r = 0;
  for (int i = 0; i < bits; i++)
    r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}
// plain cyclic DFT of length n (a power of two) with root rho, O(n^2): table building only
inline void dft(const uint64_t* v, int n, uint64_t rho, uint64_t q, uint64_t* out)
{
  for (int k = 0; k < n; k++) {
    const uint64_t wk = powm(rho, (uint64_t)k, q);
    uint64_t acc = 0, w = 1;
    for (int i = 0; i < n; i++) {
      acc = (uint64_t)(((u128)v[i] * w + acc) % q);
      w = mulm(w, wk, q);
    }
    out[k] = acc;
  }
}
// a usable m / q pair: the 256-th roots of unity exist modulo q (the kernels pick the Proth-form or the generic
// Montgomery product by the prime's form: QCP / QCG above)
inline bool supported(uint64_t m, uint64_t q) { return m == (uint64_t)M && (q & 1) && (q >> 60) == 0 && q > 256 && (q - 1) % 256 == 0; }

// rho256: any element of order 256 modulo q
inline uint64_t order256(uint64_t q)
{
  for (uint64_t h = 2;; h++) {
    const uint64_t r = powm(h, (q - 1) / 256, q);
    if (powm(r, 128, q) != 1)
      return r;
  }
}
// tab[TAB_WORDS] for one prime; root = the context's root of order m (zeta = root^2, src/bluestein.cpp:94-98)
inline void build_prime_table(uint64_t q, uint64_t root, uint64_t* tab)
{
  const uint64_t zeta = mulm(root, root, q);
  const uint64_t om[3] = {powm(zeta, M1, q), powm(zeta, M2, q), powm(zeta, M3, q)};
  const uint64_t rho256 = order256(q);
  const int ps[3] = {P1, P2, P3};
  const int tw_off[3] = {TW4, TW16, TW256}, fv[3] = {F_V1, F_V2, F_V3}, iv[3] = {I_V1, I_V2, I_V3};
  for (int d = 0; d < 3; d++) {
    const int p = ps[d], n = p - 1;
    int bits = 0;
    while ((1 << bits) < n)
      bits++;
    const uint64_t rho = powm(rho256, 256 / n, q);
    for (int e = 0; e < n; e++)
      tab[tw_off[d] + e] = mont(powm(rho, (uint64_t)e, q), q);
    const uint64_t ninv = powm((uint64_t)n, q - 2, q), oinv = powm(om[d], q - 2, q);
    const int ginv = cinv(GEN, p);
    uint64_t v[256], vi[256], hat[256];
    for (int c = 0; c < n; c++) {
      v[c] = powm(om[d], (uint64_t)cpowmod(ginv, c, p), q);     // omega^(g^-c)
      vi[c] = powm(oinv, (uint64_t)cpowmod(GEN, c, p), q);      // omega^(-g^c)
    }
    dft(v, n, rho, q, hat);
    for (int pp = 0; pp < n; pp++)
      tab[fv[d] + pp] = mont(mulm(hat[brev((unsigned)pp, bits)], ninv, q), q);
    dft(vi, n, rho, q, hat);
    for (int pp = 0; pp < n; pp++)
      tab[iv[d] + pp] = mont(mulm(hat[brev((unsigned)pp, bits)], ninv, q), q);
  }
  const uint64_t o2inv = powm(om[1], q - 2, q);
  for (int b2 = 0; b2 < 16; b2++) {
    const int j2 = gipow2(b2);
    for (int i2 = 1; i2 <= 16; i2++)
      tab[F_W2 + b2 * 16 + (i2 - 1)] = mont(powm(om[1], (uint64_t)(i2 * j2), q), q);
    for (int i2 = 0; i2 <= 16; i2++)
      tab[I_W2 + i2 * 16 + b2] = mont(powm(o2inv, (uint64_t)(i2 * j2), q), q);
  }
  for (int i = 0; i < 4; i++)
    tab[I_MINV + i] = mont(powm((uint64_t)M % q, q - 2, q), q);
}
// the index tables of the context (depend on m only)
inline void build_index_tables(uint16_t* pos2 /*16384*/, uint16_t* dlog3 /*257*/, uint16_t* gpow3 /*256*/)
{
  static uint16_t rank[M];
  uint16_t r = 0;
  for (int j = 0; j < M; j++)
    rank[j] = (j % P1 && j % P2 && j % P3) ? r++ : (uint16_t)0xffff;
  const int c1 = M1 * cinv(M1 % P1, P1), c2 = M2 * cinv(M2 % P2, P2), c3 = M3 * cinv(M3 % P3, P3);
  const int ginv3 = cinv(GEN, P3);
  int gi3[256];
  for (int b = 0, v = 1; b < 256; b++, v = v * ginv3 % P3)
    gi3[b] = v;
  dlog3[0] = 0;
  for (int a = 0, v = 1; a < 256; a++, v = v * GEN % P3) {
    gpow3[a] = (uint16_t)v;
    dlog3[v] = (uint16_t)a;
  }
  for (int c = 0; c < 64; c++) {
    const int j1 = c / 16 + 1, j2 = gipow2(c % 16);
    for (int t = 0; t < 16; t++)
      for (int k = 0; k < 16; k++) {
        const int j3 = gi3[t + 16 * k];
        const long j = ((long)j1 * c1 + (long)j2 * c2 + (long)j3 * c3) % M;
        pos2[(c * 16 + t) * 16 + k] = rank[j];
      }
  }
}
}  // namespace host

}  // namespace pfa
}  // namespace hx

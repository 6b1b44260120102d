// norm_r16.h -- register-tiled form of the canonical-embedding norm for N = 2^14 (the benchmark ring) and 2^15:
// the 8192-point complex transform of the "quarter" trick (norm_kernels.h, src/norms.cpp:200-262) as
// THREE radix-16 register passes of 512 threads x 16 points, the last stage as a lane exchange, and a pairing
// pass that forms every pair once.  The round-1/2 kernel (embed_norm_quarter_kernel) makes seven radix-4
// passes through LDS with 1024 threads x 2 butterflies each, nine barriers, every pass exposing an LDS
// round trip per two stages: 40 us per element on a CU against 7.5 us of LDS bandwidth.
//
// Decimation in frequency, output in bit-reversed order (only maxima are taken).  Stage with half-length
// len: (a, b) = (x[k], x[k+len]) -> x[k] = a + b, x[k+len] = (a - b) T_len(j), j = k mod len,
// T_len(j) = W^(j N / len) = wtab[j N/len]  (N = 16384, W = exp(2 pi i / 2N) ... as dif_fft_lds).
//   pass A  len = 4096, 2048, 1024, 512   positions  t + 512 k            (t < 512,  k < 16)
//   pass B  len =  256,  128,   64,  32   positions  512 b + j + 32 k     (b = t>>5, j = t&31)
//   pass C  len =   16,    8,    4,   2   positions   32 b + j +  2 k     (b = t>>1, j = t&1)
//   stage len = 1 between lanes t and t ^ 1: Z[p] = x[p] + x[p+1] (p even), x[p-1] - x[p] (p odd)
// Between the passes the real parts cross ONE padded array (index i stored at i + (i >> 5): pass C's stride-32
// block starts then fall on different banks), then the imaginary parts: 66 KiB, two workgroups per CU.
// The phase functions are plain C++ (HXD) so that tests/cpp/norm_replay.cpp runs them thread by thread
// on the CPU against the definition.
#pragma once
#include <stdint.h>

#ifndef HXD
#if defined(__HIPCC__)
#define HXD __host__ __device__ __forceinline__
#else
#define HXD inline
#endif
#endif

namespace hx {

struct cplx16 {
  double x, y;
};
struct tw16 {   // layout-compatible with double2
  double x, y;
};
constexpr unsigned R16_LOGN = 14, R16_N = 1u << R16_LOGN, R16_M = R16_N >> 1, R16_THREADS = 512;
constexpr unsigned R16_LDS_DOUBLES = R16_M + (R16_M >> 5);   // one of the two arrays (re / im), padded
HXD unsigned r16_pad(unsigned i) { return i + (i >> 5); }

HXD tw16 r16_cmul(tw16 a, tw16 b) { return tw16{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// four DIF stages on 16 points S apart (S = 2^LOGS): v[k] <-> position base + j0 + S k, j0 < S.
// LOGTAB: log2 of the twiddle table's size (T_len(j) = wtab[j 2^LOGTAB / len]): 14 for N = 2^14; 15 when the
// 8192-point transform is one of the two sub-transforms of N = 2^15 (embed_norm_r16_split_kernel).
// Twiddles: the stage with len = S h multiplies position j0 + S k' (k' < h) by
//   T = W^((j0 + S k') 2^LOGTAB / (S h)) = B_h * exp(pi i k' / h),   B_h = W^(j0 2^LOGTAB / (S h)),
// so ONE table entry per pass and thread is loaded (B_8; B_4 = B_8^2, B_2 = B_4^2, B_1 = B_2^2) and the
// rest are its products with the 16th roots of unity exp(pi i k / 8), read from the table at wave-uniform
// addresses (wtab[k 2^LOGTAB / 8]) -- the first version loaded all 15 twiddles of a pass per thread from the
// strided table and was latency-bound on them (slower than the LDS-pass kernel it was to replace).
template <int LOGS, int LOGTAB = 14>
HXD void r16_pass(cplx16 (&v)[16], unsigned j0, const tw16* wtab)
{
  constexpr unsigned S = 1u << LOGS;
  tw16 B = wtab[j0 * ((1u << LOGTAB) / (S * 8u))];
#pragma unroll
  for (unsigned h = 8; h >= 1; h >>= 1) {
#pragma unroll
    for (unsigned k = 0; k < 16; k++) {
      if (k & h)
        continue;
      const unsigned kp = k & (h - 1);
      // exp(pi i kp / h) = W^(kp 2^LOGTAB / h)
      const tw16 T = kp == 0 ? B : r16_cmul(B, wtab[kp * ((1u << LOGTAB) / h)]);
      const cplx16 a = v[k], b = v[k + h];
      const double dx = a.x - b.x, dy = a.y - b.y;
      v[k].x = a.x + b.x;
      v[k].y = a.y + b.y;
      v[k + h].x = dx * T.x - dy * T.y;
      v[k + h].y = dx * T.y + dy * T.x;
    }
    B = r16_cmul(B, B);
  }
}
// where thread t keeps its 16 points of each pass: position of v[k]
HXD unsigned r16_pos_A(unsigned t, unsigned k) { return t + 512u * k; }
HXD unsigned r16_pos_B(unsigned t, unsigned k) { return 512u * (t >> 5) + (t & 31u) + 32u * k; }
HXD unsigned r16_pos_C(unsigned t, unsigned k) { return 32u * (t >> 1) + (t & 1u) + 2u * k; }

// max of |f|^2 at the two evaluation points that Z_j = z and Z_(M-1-j) = partner give, w = W^(2j+1)
// (embed_norm_quarter_kernel's formula)
HXD double r16_pair_norm2(cplx16 z, cplx16 partner, tw16 w)
{
  const double cr = partner.x, ci = -partner.y;
  const double er = 0.5 * (z.x + cr), ei = 0.5 * (z.y + ci);
  const double dr = z.x - cr, di = z.y - ci;
  const double orr = 0.5 * di, oi = -0.5 * dr;
  const double tr = orr * w.x - oi * w.y, ti = orr * w.y + oi * w.x;
  const double a = (er + tr) * (er + tr) + (ei + ti) * (ei + ti);
  const double b = (er - tr) * (er - tr) + (ei - ti) * (ei - ti);
  return a > b ? a : b;
}
HXD unsigned r16_brev4(unsigned i) { return ((i & 1u) << 3) | ((i & 2u) << 1) | ((i & 4u) >> 1) | ((i & 8u) >> 3); }
// ---- the last stage and the pairing ----
// After pass C thread t holds
// positions p = 32 (t>>1) + (t&1) + 2k: the partner p ^ 1 of the last stage (len = 1) is the same k of lane t ^ 1,
// so that stage is a lane exchange and the finished Z stay in registers:
HXD cplx16 r16_last_lane(cplx16 own, cplx16 other, unsigned t)   // other: the value lane t ^ 1 holds at the same k
{
  return (t & 1u) ? cplx16{other.x - own.x, other.y - own.y} : cplx16{own.x + other.x, own.y + other.y};
}
// Pairing: Z at p meets Z at M - 1 - p = pos_C(511 - t, 15 - k); (p, M-1-p) and (M-1-p, p) give the same two
// evaluation points, so thread t pairs its k < 8 with what thread 511 - t holds at 15 - k >= 8 -- every unordered
// pair once (the two-array kernel formed each twice).  The upper halves cross through the array at
// [kk][t], kk = k - 8 (real parts; imaginary parts R16_XCHG_IM further on).
constexpr unsigned R16_XCHG_IM = 4096;
HXD unsigned r16_xchg_idx(unsigned t, unsigned kk) { return kk * 512u + t; }
HXD unsigned r16_brev8(unsigned x)
{
  unsigned r = 0;
  for (int i = 0; i < 8; i++)
    r |= ((x >> i) & 1u) << (7 - i);
  return r;
}
// j = brev13(p) = 4096 (t&1) + 256 brev4(k) + brev8(t>>1):  W^(2j+1) = wtab[r16_pair_tw_thread(t)] * wtab[r16_pair_tw_k(k)]
HXD unsigned r16_pair_tw_thread(unsigned t) { return 8192u * (t & 1u) + 2u * r16_brev8(t >> 1) + 1u; }
HXD unsigned r16_pair_tw_k(unsigned k) { return 512u * r16_brev4(k); }

// N = 2^15 as S = 2 sub-transforms of H = 8192 points (norm_kernels.h, embed_norm_quarter_split_kernel):
// input point i of sub-transform `sub`: h_i = sum_{t<2} z_(i+tH) U^((i+tH) sub),
//   z_n U^(n sub) = (f_2n + i f_(2n+1)) W^(2n (2 sub + 1)),  W = exp(2 pi i / 2N),  wtab[k] = W^k for k < N
HXD cplx16 r16_split_input(const double* f, const tw16* wtab, unsigned i, unsigned sub)
{
  constexpr unsigned N = 1u << 15, H = 8192u, mmask = 2u * N - 1u;
  cplx16 acc{0.0, 0.0};
  for (unsigned t = 0; t < 2; t++) {
    const unsigned idx = i + t * H;
    const unsigned e = (2u * idx * (2u * sub + 1u)) & mmask;
    const tw16 w = wtab[e & (N - 1u)];
    const double vx = f[2 * idx], vy = f[2 * idx + 1];
    const double zr = vx * w.x - vy * w.y, zi = vx * w.y + vy * w.x;
    acc.x += e >= N ? -zr : zr;
    acc.y += e >= N ? -zi : zi;
  }
  return acc;
}
// N = 2^15 with both sub-transforms at once (embed_norm_r16x2_kernel): 1024 threads, half h = sub-transform h in its
// own array; after the lane-exchange stage Z_h sits in registers at pos_C(t, k).  Z0 at p meets Z1 at 8191 - p, i.e.
// what thread 511 - t of the OTHER half holds at 15 - k: half 0 forms the pairs of its k < 8, half 1 those of its
// own k < 8 (the pairs of half 0's k >= 8) -- every pair once, the upper halves crossing through the arrays as in the
// N = 2^14 kernel.  Twiddle W^(4 brev13(p) + 1) for the Z0 position p of the pair (W = exp(2 pi i / 2^16)):
//   half 0, own p:        4 j + 1,       j = 4096 (t&1) + 256 brev4(k) + brev8(t>>1)
//   half 1, p = 8191 - p': 32765 - 4 j'  (j' from its own t, k) = thread part * conj(W^(1024 brev4(k)))
HXD unsigned r16x2_pair_tw_thread(unsigned h, unsigned t)
{
  const unsigned a = 16384u * (t & 1u) + 4u * r16_brev8(t >> 1);
  return h ? 32765u - a : a + 1u;
}
HXD unsigned r16x2_pair_tw_k(unsigned k) { return 1024u * r16_brev4(k); }
HXD double r16x2_pair(unsigned h, cplx16 own, cplx16 other, tw16 wth, tw16 wk, unsigned k)
{
  const tw16 u = h ? tw16{wk.x, -wk.y} : wk;
  const tw16 w = k == 0 ? wth : r16_cmul(wth, u);
  return h ? r16_pair_norm2(other, own, w) : r16_pair_norm2(own, other, w);
}

}  // namespace hx

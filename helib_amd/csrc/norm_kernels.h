// norm_kernels.h -- canonical-embedding norm of real-coefficient polynomials on the device
// (SURVEY "next" row N1): embeddingLargestCoeff (src/norms.cpp:129-262, 480-493) =
//     max over j in Z_m^* of | f(W^j) |,   W = exp(2*pi*i/m),
// which the reference evaluates with PGFFT (a complex-double DFT of size m, m/2 or m/4) and uses
// to measure the noise added by Ctxt::modDownToSet (src/Ctxt.cpp:466-530) and
// DoubleCRT::breakIntoDigits (src/DoubleCRT.cpp:538-545).  Floating point: parity with the
// reference is to a relative tolerance (tests/GTestPGFFT.cpp:299-303 uses 1e-9-class bounds).
//
// m = 2N a power of two ("odd-power trick", src/norms.cpp:159-198): with g_i = f_i W^i,
// f(W^(2j+1)) = sum_i g_i V^(ij), V = W^2 -- one N-point complex DFT.  Only the maximum modulus is
// wanted, so the output order is irrelevant: decimation-in-frequency, in place, no bit-reversal,
// two stages per LDS pass.  N <= 2^14 uses the real-input "quarter" form (one N/2-point transform
// in one workgroup); larger N (up to 2^17) the same form split into S = N/2/8192 sub-transforms, one
// workgroup per pair of them (embed_norm_quarter_split_kernel); embed_norm_kernel is the plain
// N-point form, split by log2(S) DIF levels applied while loading into S independent H-point
// transforms (H = N/S <= 8192 complex doubles = 128 KiB of LDS), one workgroup each.  Every workgroup folds its maximum into out2[row] with an atomic max on the bit pattern
// (non-negative doubles order like unsigned integers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "norm_r16.h"

namespace hx {

constexpr int NORM_MAX_LOGH = 13;  // 2^13 complex doubles = 128 KiB LDS
constexpr int NORM_THREADS = 1024;

struct cplx {
  double x, y;
};
__device__ __forceinline__ cplx cmul(cplx a, double2 w) { return {a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cplx cmul_i(cplx a) { return {-a.y, a.x}; }

// In-place decimation-in-frequency DFT of the H = 2^logh points in (re, im).  Both callers use a
// root for which the stage of half-length len multiplies position j by W^(j*N/len) = wtab[j*N/len]
// (N = table size): the H-point root is W^(2N/H).
// Two stages are fused into one radix-4 pass (one barrier, one LDS round trip per two stages):
//   stage len  : x0=a0+a2, x2=(a0-a2)T, x1=a1+a3, x3=(a1-a3)T*i      (T = T_len(j), T_len(j+len/2) = i T)
//   stage len/2: y0=x0+x1, y1=(x0-x1)T2, y2=x2+x3, y3=(x2-x3)T2     (T2 = T_(len/2)(j))
// Output order is bit-reversed; callers only take maxima (or pair positions p and H-1-p).
// CLOGH / CNTH: compile-time transform size and thread count (0 = take the runtime arguments).  With
// both known the loops over a thread's butterflies have constant trip counts and are unrolled, so that
// the twiddle and LDS loads of all of a thread's butterflies in a pass are in flight together -- with
// runtime bounds every iteration exposed its own global-memory round trip (the norm kernels were
// latency-bound: 40 us per 8192-point transform, one workgroup per CU).
template <int CLOGH = 0, int CNTH = 0>
__device__ __forceinline__ void dif_fft_lds(double* re, double* im, int logh_rt, unsigned tw_half,
                                            const double2* __restrict__ wtab, unsigned tid, unsigned nth_rt)
{
  // tw_half: the table size N
  const int logh = CLOGH ? CLOGH : logh_rt;
  const unsigned nth = CNTH ? (unsigned)CNTH : nth_rt;
  const unsigned H = 1u << logh;
  int stages = logh;
  unsigned len = H >> 1;
#pragma unroll
  while (stages >= 2) {
    const unsigned hl = len >> 1;        // j < len/2
    const unsigned s1 = tw_half / len;   // T_len(j)      = wtab[j * s1]
    const unsigned s2 = s1 * 2;          // T_(len/2)(j)  = wtab[j * s2]
#pragma unroll
    for (unsigned q = tid; q < (H >> 2); q += nth) {
      const unsigned j = q & (hl - 1), blk = q / hl, k = blk * 2 * len + j;
      const cplx a0{re[k], im[k]}, a1{re[k + hl], im[k + hl]}, a2{re[k + len], im[k + len]},
          a3{re[k + len + hl], im[k + len + hl]};
      const double2 T = wtab[j * s1], T2 = wtab[j * s2];
      const cplx x0 = cadd(a0, a2), x2 = cmul(csub(a0, a2), T);
      const cplx x1 = cadd(a1, a3), x3 = cmul_i(cmul(csub(a1, a3), T));
      const cplx y0 = cadd(x0, x1), y1 = cmul(csub(x0, x1), T2);
      const cplx y2 = cadd(x2, x3), y3 = cmul(csub(x2, x3), T2);
      re[k] = y0.x, im[k] = y0.y;
      re[k + hl] = y1.x, im[k + hl] = y1.y;
      re[k + len] = y2.x, im[k + len] = y2.y;
      re[k + len + hl] = y3.x, im[k + len + hl] = y3.y;
    }
    __syncthreads();
    len >>= 2;
    stages -= 2;
  }
  if (stages == 1) {  // len == 1: twiddle 1
#pragma unroll
    for (unsigned k2 = tid; k2 < (H >> 1); k2 += nth) {
      const unsigned k = 2 * k2;
      const double ar = re[k], ai = im[k], br = re[k + 1], bi = im[k + 1];
      re[k] = ar + br, im[k] = ai + bi;
      re[k + 1] = ar - br, im[k + 1] = ai - bi;
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void block_max_to(double mx, double* sm, unsigned tid, unsigned nth,
                                             unsigned long long* dst, bool direct = false)
{
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(mx, off, 64);
    mx = o > mx ? o : mx;
  }
  __syncthreads();
  if ((tid & 63u) == 0)
    sm[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    const unsigned nw = (nth + 63) >> 6;
    for (unsigned w = 1; w < nw; w++)
      mx = sm[w] > mx ? sm[w] : mx;
    // direct: this workgroup is the only writer of *dst (one workgroup per polynomial) -- a plain store, which may go
    // to device-visible host memory; otherwise several workgroups meet in a zeroed word
    if (direct)
      *dst = (unsigned long long)__double_as_longlong(mx);
    else
      atomicMax(dst, (unsigned long long)__double_as_longlong(mx));
  }
}

// f: rows x N doubles; wtab[k] = W^k for k < N (W^(k+N) = -W^k); out2[row] must be zeroed.
// General form: the N-point DFT of g_i = f_i W^i, S = N/H sub-transforms per row.
__global__ void __launch_bounds__(NORM_THREADS)
embed_norm_kernel(const double* __restrict__ f, const double2* __restrict__ wtab, int logn, int logh,
                  unsigned long long* __restrict__ out2)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const unsigned N = 1u << logn, H = 1u << logh, S = N >> logh;
  double* re = sm;
  double* im = sm + H;
  const unsigned row = blockIdx.x / S, s = blockIdx.x % S;
  const unsigned tid = threadIdx.x, nth = blockDim.x;
  const double* fr = f + (size_t)row * N;
  const unsigned mmask = 2 * N - 1;
  // load: h_i = sum_t f_(i+tH) W^((i+tH)(2s+1))   (outputs j = s mod S of the N-point DFT)
  for (unsigned i = tid; i < H; i += nth) {
    double ar = 0, ai = 0;
    for (unsigned t = 0; t < S; t++) {
      const unsigned idx = i + t * H;
      const unsigned e = (idx * (2 * s + 1)) & mmask;
      double2 w = wtab[e & (N - 1)];
      const double x = e >= N ? -fr[idx] : fr[idx];
      ar += x * w.x;
      ai += x * w.y;
    }
    re[i] = ar;
    im[i] = ai;
  }
  __syncthreads();
  // root U = W^(2S) (primitive H-th root): T_len(j) = U^(j H/(2 len)) = W^(j N/len)
  dif_fft_lds(re, im, logh, N, wtab, tid, nth);
  double mx = 0;
  for (unsigned i = tid; i < H; i += nth) {
    const double n2 = re[i] * re[i] + im[i] * im[i];
    mx = n2 > mx ? n2 : mx;
  }
  block_max_to(mx, sm, tid, nth, out2 + row);
}

// "Quarter" form for N <= 2^14 (src/norms.cpp:200-262 has the reference's version of the trick):
// with e_i = f_2i, o_i = f_(2i+1), M = N/2 and V = W^2,
//   f(W^(2j+1)) = E_j + W^(2j+1) O_j,  f(W^(2(j+M)+1)) = E_j - W^(2j+1) O_j,
//   E_j = sum_i e_i V^(i(2j+1)),  conj(E_j) = E_(M-1-j)  (real e), likewise O,
// so ONE M-point complex DFT of z_i = (e_i + i o_i) V^i gives Z_j = E_j + i O_j and
//   E_j = (Z_j + conj Z_(M-1-j))/2,  O_j = (Z_j - conj Z_(M-1-j))/(2i).
// In the bit-reversed output order Z_j sits at p = brev(j) and Z_(M-1-j) at M-1-p.
// Coefficient sources: doubles in memory, or the (x, S) pair of the fused single-prime mod-down
// (delta/qd = x/qd - S) read directly -- no fdelta array is materialised for the norm.
struct NormSrcF64 {
  const double* f;
  __device__ __forceinline__ double2 pair(unsigned row, unsigned N, unsigned i) const
  {
    return reinterpret_cast<const double2*>(f + (size_t)row * N)[i];
  }
};
struct NormSrcXS {
  const uint64_t* xs;
  const int64_t* S;
  double inv_qd;
  __device__ __forceinline__ double2 pair(unsigned row, unsigned N, unsigned i) const
  {
    const ulonglong2 x = reinterpret_cast<const ulonglong2*>(xs + (size_t)row * N)[i];
    const longlong2 s = reinterpret_cast<const longlong2*>(S + (size_t)row * N)[i];
    return make_double2((double)x.x * inv_qd - (double)s.x, (double)x.y * inv_qd - (double)s.y);
  }
};

// CLOGN: compile-time log2 N for the sizes that matter (13, 14; launched with NORM_THREADS threads),
// 0 = any size from the runtime argument
template <class SRC, int CLOGN = 0>
__global__ void __launch_bounds__(NORM_THREADS)
embed_norm_quarter_kernel(SRC src, const double2* __restrict__ wtab, int logn_rt,
                          unsigned long long* __restrict__ out2)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int logn = CLOGN ? CLOGN : logn_rt;
  const unsigned N = 1u << logn, M = N >> 1;
  const int logm = logn - 1;
  double* re = sm;
  double* im = sm + M;
  const unsigned row = blockIdx.x, tid = threadIdx.x, nth = CLOGN ? (unsigned)NORM_THREADS : blockDim.x;
#pragma unroll
  for (unsigned i = tid; i < M; i += nth) {
    const double2 v = src.pair(row, N, i);   // (f_2i, f_(2i+1))
    const double2 w = wtab[2 * i];    // V^i = W^(2i)
    re[i] = v.x * w.x - v.y * w.y;
    im[i] = v.x * w.y + v.y * w.x;
  }
  __syncthreads();
  // root V^2 = W^4 (primitive M-th root): T_len(j) = W^(4 j M/(2 len)) = W^(j N/len)
  dif_fft_lds<(CLOGN ? CLOGN - 1 : 0), (CLOGN ? NORM_THREADS : 0)>(re, im, logm, N, wtab, tid, nth);
  double mx = 0;
#pragma unroll
  for (unsigned p = tid; p < M; p += nth) {
    const unsigned j = __brev(p) >> (32 - logm);
    const double zr = re[p], zi = im[p], cr = re[M - 1 - p], ci = -im[M - 1 - p];
    const double er = 0.5 * (zr + cr), ei = 0.5 * (zi + ci);
    // O = (Z - C)/(2i) = (-i/2)(Z - C)
    const double dr = zr - cr, di = zi - ci;
    const double orr = 0.5 * di, oi = -0.5 * dr;
    const double2 w = wtab[2 * j + 1];
    const double tr = orr * w.x - oi * w.y, ti = orr * w.y + oi * w.x;
    const double a = (er + tr) * (er + tr) + (ei + ti) * (ei + ti);
    const double b = (er - tr) * (er - tr) + (ei - ti) * (ei - ti);
    const double n2 = a > b ? a : b;
    mx = n2 > mx ? n2 : mx;
  }
  block_max_to(mx, sm, tid, nth, out2 + row);
}

// N = 2^14 (the benchmark ring), register-tiled: norm_r16.h.  512 threads x 16 points, three radix-16 register
// passes, four stages per LDS round trip, 16 independent butterflies per thread in flight.
//
// Round 4: ONE padded array (66 KiB) instead of one per component, so that TWO workgroups share a CU.  The round-3
// form (132 KiB) ran one workgroup per CU: all 256 of a launch load their rows at the same time (262 KB each: the
// launch's 67-134 MB in one burst, ~10 us), then all compute with the memory system idle (~12 us), then the next
// round of rows does the same -- 384 rows cost as much as 512.  With two co-resident workgroups one computes while
// the other loads, and every row of a launch of <= 512 rows is resident at once.  The price is barriers (the real
// parts cross the array, then the imaginary parts); the last stage moved from the pairing pass into a lane
// exchange (DPP quad_perm), and the pairing forms each pair once instead of twice.
__device__ __forceinline__ double lane_xor1(double v)
{
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// FROM -> TO: positions of the 16 values before / after (r16_pos_A / _B / _C)
template <unsigned (*FROM)(unsigned, unsigned), unsigned (*TO)(unsigned, unsigned)>
__device__ __forceinline__ void r16_transpose(cplx16 (&v)[16], double* sm, unsigned t)
{
#pragma unroll
  for (unsigned k = 0; k < 16; k++)
    sm[r16_pad(FROM(t, k))] = v[k].x;
  __syncthreads();
#pragma unroll
  for (unsigned k = 0; k < 16; k++)
    v[k].x = sm[r16_pad(TO(t, k))];
  __syncthreads();
#pragma unroll
  for (unsigned k = 0; k < 16; k++)
    sm[r16_pad(FROM(t, k))] = v[k].y;
  __syncthreads();
#pragma unroll
  for (unsigned k = 0; k < 16; k++)
    v[k].y = sm[r16_pad(TO(t, k))];
}
template <class SRC>
__global__ void __launch_bounds__(R16_THREADS)
embed_norm_r16_kernel(SRC src, const double2* __restrict__ wtab, unsigned long long* __restrict__ out2, bool direct)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const unsigned row = blockIdx.x, t = threadIdx.x;
  const tw16* wt = reinterpret_cast<const tw16*>(wtab);
  cplx16 v[16];
#pragma unroll
  for (unsigned k = 0; k < 16; k++) {
    const unsigned i = r16_pos_A(t, k);
    const double2 p = src.pair(row, R16_N, i);   // (f_2i, f_(2i+1))
    // V^i = W^(2i) = W^(2t) W^(1024 k): one table entry per thread, the 16 constants at uniform addresses
    const tw16 w = k == 0 ? wt[2 * t] : r16_cmul(wt[2 * t], wt[1024u * k]);
    v[k].x = p.x * w.x - p.y * w.y;
    v[k].y = p.x * w.y + p.y * w.x;
  }
  r16_pass<9>(v, t, wt);
  r16_transpose<r16_pos_A, r16_pos_B>(v, sm, t);
  r16_pass<5>(v, t & 31u, wt);
  __syncthreads();
  r16_transpose<r16_pos_B, r16_pos_C>(v, sm, t);
  r16_pass<1>(v, t & 1u, wt);
#pragma unroll
  for (unsigned k = 0; k < 16; k++)
    v[k] = r16_last_lane(v[k], cplx16{lane_xor1(v[k].x), lane_xor1(v[k].y)}, t);
  __syncthreads();
#pragma unroll
  for (unsigned kk = 0; kk < 8; kk++) {
    sm[r16_xchg_idx(t, kk)] = v[8 + kk].x;
    sm[R16_XCHG_IM + r16_xchg_idx(t, kk)] = v[8 + kk].y;
  }
  __syncthreads();
  double mx = 0;
  const tw16 wth = wt[r16_pair_tw_thread(t)];
#pragma unroll
  for (unsigned k = 0; k < 8; k++) {
    const unsigned o = r16_xchg_idx(R16_THREADS - 1u - t, 7u - k);
    const cplx16 partner{sm[o], sm[R16_XCHG_IM + o]};
    const tw16 w = k == 0 ? wth : r16_cmul(wth, wt[r16_pair_tw_k(k)]);
    const double n2 = r16_pair_norm2(v[k], partner, w);
    mx = n2 > mx ? n2 : mx;
  }
  block_max_to(mx, sm, t, R16_THREADS, out2 + row, direct);
}

// N = 2^15, round 4: BOTH sub-transforms at once -- 1024 threads, half h = sub-transform h in its own 66 KiB array
// (norm_r16.h: r16x2_*), last stage as a lane exchange, the pairing across the halves through the arrays; nothing
// parked in global memory.  The round-2 kernel below (radix-4 LDS passes, sub-transform 1 parked, then sub-transform
// 0) takes 73 us for the 192 polynomials of a CKKS multiply, one workgroup each on 192 of the 256 CUs.
__global__ void __launch_bounds__(2 * R16_THREADS)
embed_norm_r16x2_kernel(const double* __restrict__ f, const double2* __restrict__ wtab, unsigned long long* __restrict__ out2,
                        bool direct)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const unsigned row = blockIdx.x, h = threadIdx.x >> 9, t = threadIdx.x & 511u;
  double* mine = sm + h * R16_LDS_DOUBLES;
  const double* theirs = sm + (1u - h) * R16_LDS_DOUBLES;
  const tw16* wt = reinterpret_cast<const tw16*>(wtab);
  const double* fr = f + (size_t)row * (1u << 15);
  cplx16 v[16];
#pragma unroll
  for (unsigned k = 0; k < 16; k++)
    v[k] = r16_split_input(fr, wt, r16_pos_A(t, k), h);
  r16_pass<9, 15>(v, t, wt);
  r16_transpose<r16_pos_A, r16_pos_B>(v, mine, t);
  r16_pass<5, 15>(v, t & 31u, wt);
  __syncthreads();
  r16_transpose<r16_pos_B, r16_pos_C>(v, mine, t);
  r16_pass<1, 15>(v, t & 1u, wt);
#pragma unroll
  for (unsigned k = 0; k < 16; k++)
    v[k] = r16_last_lane(v[k], cplx16{lane_xor1(v[k].x), lane_xor1(v[k].y)}, t);
  __syncthreads();
#pragma unroll
  for (unsigned kk = 0; kk < 8; kk++) {
    mine[r16_xchg_idx(t, kk)] = v[8 + kk].x;
    mine[R16_XCHG_IM + r16_xchg_idx(t, kk)] = v[8 + kk].y;
  }
  __syncthreads();
  double mx = 0;
  const tw16 wth = wt[r16x2_pair_tw_thread(h, t)];
#pragma unroll
  for (unsigned k = 0; k < 8; k++) {
    const unsigned o = r16_xchg_idx(R16_THREADS - 1u - t, 7u - k);
    const cplx16 other{theirs[o], theirs[R16_XCHG_IM + o]};
    const double n2 = r16x2_pair(h, v[k], other, wth, wt[r16x2_pair_tw_k(k)], k);
    mx = n2 > mx ? n2 : mx;
  }
  block_max_to(mx, sm, threadIdx.x, 2 * R16_THREADS, out2 + row, direct);
}

// The quarter form for N > 2^14 (M = N/2 = S*H points, H = 8192, S = 2, 4, 8): the M-point transform
// of z_i = (f_2i + i f_(2i+1)) V^i as S sub-transforms of H points, the first log2(S) decimation
// levels applied while loading (sub-transform s holds Z_j for j = s + S k', at p = brev(k')).  The
// pairing partner Z_(M-1-j) lives in sub-transform S-1-s at position H-1-p, so one workgroup takes
// the pair (s, S-1-s) of one row: sub-transform S-1-s first, parked in global memory (park:
// [row][s][H] complex), then sub-transform s in LDS and the pairing pass of the quarter kernel over
// its H positions -- the positions of sub-transform S-1-s give the complex conjugates of the same
// four values.  Half the transform work and half the reads of embed_norm_kernel at the same N.
__global__ void __launch_bounds__(NORM_THREADS)
embed_norm_quarter_split_kernel(const double* __restrict__ f, const double2* __restrict__ wtab, int logn, int logh,
                                double2* __restrict__ park, unsigned long long* __restrict__ out2)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const unsigned N = 1u << logn, M = N >> 1, H = 1u << logh, S = M >> logh, half = S >> 1;
  double* re = sm;
  double* im = sm + H;
  const unsigned row = blockIdx.x / half, s = blockIdx.x % half;
  const unsigned tid = threadIdx.x, nth = blockDim.x;
  const double2* fp = reinterpret_cast<const double2*>(f + (size_t)row * N);
  double2* pk = park + ((size_t)row * half + s) * H;
  const unsigned mmask = 2 * N - 1;
  for (int pass = 0; pass < 2; pass++) {
    const unsigned sub = pass == 0 ? S - 1 - s : s;
    // h_i = sum_t z_(i+tH) U^((i+tH) sub),  z_n U^(n sub) = (f_2n + i f_(2n+1)) W^(2n (2 sub + 1))
    for (unsigned i = tid; i < H; i += nth) {
      double ar = 0, ai = 0;
      for (unsigned t = 0; t < S; t++) {
        const unsigned idx = i + t * H;
        const unsigned e = (2u * idx * (2u * sub + 1u)) & mmask;
        const double2 w = wtab[e & (N - 1)];
        const double2 v = fp[idx];
        const double zr = v.x * w.x - v.y * w.y, zi = v.x * w.y + v.y * w.x;
        ar += e >= N ? -zr : zr;
        ai += e >= N ? -zi : zi;
      }
      re[i] = ar;
      im[i] = ai;
    }
    __syncthreads();
    // root U^S = W^(4S) = W^(2N/H): the table stride dif_fft_lds expects (H = 8192, 1024 threads: always)
    dif_fft_lds<NORM_MAX_LOGH, NORM_THREADS>(re, im, logh, N, wtab, tid, nth);
    if (pass == 0) {
      for (unsigned p = tid; p < H; p += nth)
        pk[p] = make_double2(re[p], im[p]);
      __syncthreads();
    }
  }
  double mx = 0;
  for (unsigned p = tid; p < H; p += nth) {
    const unsigned j = s + S * (__brev(p) >> (32 - logh));
    const double2 c = pk[H - 1 - p];
    const double zr = re[p], zi = im[p], cr = c.x, ci = -c.y;
    const double er = 0.5 * (zr + cr), ei = 0.5 * (zi + ci);
    const double dr = zr - cr, di = zi - ci;
    const double orr = 0.5 * di, oi = -0.5 * dr;
    const double2 w = wtab[2 * j + 1];
    const double tr = orr * w.x - oi * w.y, ti = orr * w.y + oi * w.x;
    const double a = (er + tr) * (er + tr) + (ei + ti) * (ei + ti);
    const double b = (er - tr) * (er - tr) + (ei - ti) * (ei - ti);
    const double n2 = a > b ? a : b;
    mx = n2 > mx ? n2 : mx;
  }
  block_max_to(mx, sm, tid, nth, out2 + row);
}

// =====================================================================
// General m: Bluestein in complex double (the reference's PGFFT does the same for a non-power-of-two
// size, src/PGFFT.cpp).  With omega = exp(2 pi i/m) and v_k = exp(pi i k^2/m),
//     f(omega^j) = v_j * sum_i (f_i v_i) conj(v_(j-i)),
// so |f(omega^j)| = |conv_j| for the cyclic convolution of a_i = f_i v_i with c_k = conj(v_|k|) over
// P = 2^bk >= 2m-1 points (the same size as the integer Bluestein transform).  P-point transforms are
// S = P/H sub-transforms of H <= 8192 points held in LDS (one workgroup each), the first log2(S)
// decimation-in-frequency levels folded into the load and, for the inverse, into the final
// gather over the units of Z_m^*:
//   forward   X[s + S k'] = sum_{i<H} ( sum_{t<S} x[i+tH] W^((i+tH)s) ) W_H^(i k'),  W = exp(2 pi i/P)
//   inverse   x[i + tH]   = (1/P) sum_{s<S} W^(-(i+tH)s) y_s[i],   y_s[i] = sum_k' X[s+S k'] W_H^(-i k')
// Transform-domain data is kept as [row][s][p] with p the bit-reversed k' (dif_fft_lds order).
// =====================================================================
// inverse of dif_fft_lds without the 1/H: bit-reversed input, natural output, conjugate twiddles
__device__ __forceinline__ void dit_ifft_lds(double* re, double* im, int logh, unsigned tw_half,
                                             const double2* __restrict__ wtab, unsigned tid, unsigned nth)
{
  const unsigned H = 1u << logh;
  for (unsigned len = 1; len < H; len <<= 1) {
    const unsigned s1 = tw_half / len;
    for (unsigned q = tid; q < (H >> 1); q += nth) {
      const unsigned j = q & (len - 1), k = ((q / len) * 2 * len) + j;
      const double2 T = wtab[j * s1];
      const double br = re[k + len] * T.x + im[k + len] * T.y;   // B * conj(T)
      const double bi = im[k + len] * T.x - re[k + len] * T.y;
      const double ar = re[k], ai = im[k];
      re[k] = ar + br, im[k] = ai + bi;
      re[k + len] = ar - br, im[k + len] = ai - bi;
    }
    __syncthreads();
  }
}
__device__ __forceinline__ double2 wpow(const double2* __restrict__ wtab, unsigned e, unsigned P)
{
  e &= P - 1;
  const unsigned half = P >> 1;
  double2 w = wtab[e & (half - 1)];
  if (e >= half) {
    w.x = -w.x;
    w.y = -w.y;
  }
  return w;
}
// forward transform of one (row, s) block.  mode 0: x[n] = f[row][n] * v[n] (n < phim), Z multiplied by
// chat; mode 1 (set-up): x = the complex vector cin (P entries), Z stored as is.
__global__ void __launch_bounds__(NORM_THREADS)
bnorm_fwd_kernel(const double* __restrict__ f, const double2* __restrict__ v, const double2* __restrict__ cin,
                 const double2* __restrict__ wtab, const double2* __restrict__ chat,
                 double2* __restrict__ Z, int logp, int logh, unsigned phim, int mode)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const unsigned P = 1u << logp, H = 1u << logh, S = P >> logh;
  double* re = sm;
  double* im = sm + H;
  const unsigned row = blockIdx.x / S, s = blockIdx.x % S;
  const unsigned tid = threadIdx.x, nth = blockDim.x;
  for (unsigned i = tid; i < H; i += nth) {
    double ar = 0, ai = 0;
    for (unsigned t = 0; t < S; t++) {
      const unsigned n = i + t * H;
      double xr, xi;
      if (mode == 0) {
        if (n >= phim)
          break;
        const double fv = f[(size_t)row * phim + n];
        const double2 vv = v[n];
        xr = fv * vv.x;
        xi = fv * vv.y;
      } else {
        const double2 cv = cin[n];
        xr = cv.x;
        xi = cv.y;
      }
      const double2 w = wpow(wtab, n * s, P);
      ar += xr * w.x - xi * w.y;
      ai += xr * w.y + xi * w.x;
    }
    re[i] = ar;
    im[i] = ai;
  }
  __syncthreads();
  dif_fft_lds(re, im, logh, P >> 1, wtab, tid, nth);
  double2* out = Z + ((size_t)row * S + s) * H;
  for (unsigned p = tid; p < H; p += nth) {
    double zr = re[p], zi = im[p];
    if (chat) {
      const double2 c = chat[(size_t)s * H + p];
      const double t = zr * c.x - zi * c.y;
      zi = zr * c.y + zi * c.x;
      zr = t;
    }
    out[p] = make_double2(zr, zi);
  }
}
// y_s = H-point inverse sub-transform of block (row, s), in place
__global__ void __launch_bounds__(NORM_THREADS)
bnorm_inv_kernel(double2* __restrict__ Z, const double2* __restrict__ wtab, int logp, int logh)
{
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const unsigned P = 1u << logp, H = 1u << logh;
  double* re = sm;
  double* im = sm + H;
  const unsigned tid = threadIdx.x, nth = blockDim.x;
  double2* blk = Z + (size_t)blockIdx.x * H;
  for (unsigned p = tid; p < H; p += nth) {
    const double2 z = blk[p];
    re[p] = z.x;
    im[p] = z.y;
  }
  __syncthreads();
  dit_ifft_lds(re, im, logh, P >> 1, wtab, tid, nth);
  for (unsigned i = tid; i < H; i += nth)
    blk[i] = make_double2(re[i], im[i]);
}
// out2[row] = max over the units n of Z_m^* of |conv_n|^2 / P^2,  conv_n = sum_s W^(-n s) y_s[n mod H]
__global__ void __launch_bounds__(256)
bnorm_max_kernel(const double2* __restrict__ Y, const double2* __restrict__ wtab, const uint32_t* __restrict__ zms,
                 unsigned phim, int logp, int logh, unsigned long long* __restrict__ out2)
{
  __shared__ double smax[8];
  const unsigned P = 1u << logp, H = 1u << logh, S = P >> logh;
  const unsigned row = blockIdx.y, tid = threadIdx.x;
  const double2* y = Y + (size_t)row * P;
  const double inv = 1.0 / (double)P;
  double mx = 0;
  for (unsigned idx = blockIdx.x * blockDim.x + tid; idx < phim; idx += gridDim.x * blockDim.x) {
    const unsigned n = zms[idx], i = n & (H - 1);
    double cr = 0, ci = 0;
    for (unsigned s = 0; s < S; s++) {
      const double2 w = wpow(wtab, n * s, P);          // conj(w) = W^(-n s)
      const double2 yv = y[(size_t)s * H + i];
      cr += yv.x * w.x + yv.y * w.y;
      ci += yv.y * w.x - yv.x * w.y;
    }
    cr *= inv;
    ci *= inv;
    const double n2 = cr * cr + ci * ci;
    mx = n2 > mx ? n2 : mx;
  }
  block_max_to(mx, smax, tid, blockDim.x, out2 + row);
}

// fdelta of the fused single-prime scale-down: delta = x - qd*S  =>  delta/qd = x/qd - S
__global__ void __launch_bounds__(256)
frac_from_xs_kernel(const uint64_t* __restrict__ xs, const int64_t* __restrict__ S, double inv_qd,
                    double* __restrict__ out, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = (double)xs[i] * inv_qd - (double)S[i];
}

}  // namespace hx

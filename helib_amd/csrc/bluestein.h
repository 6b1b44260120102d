// bluestein.h -- device side of Cmodulus::FFT / iFFT for general (non power-of-two) m:
// HElib src/CModulus.cpp:431-443, :555-577 and src/bluestein.cpp:76-201.
//
//   forward : x_i *= powers[i]  ->  linear convolution with the chirp b  ->  window / fold
//             -> *= powers[k]   ->  keep k in Z_m^* (increasing)
//   inverse : scatter y onto Z_m^*  ->  the same with rInv  ->  rem Phi_m  ->  * m^-1
//
// The convolution (NTL fftRep in the reference) is a negacyclic NTT product of size
// 2^bk >= 2m-1 on the row kernels (conv_core.h); "rem Phi_m" (reference: zz_pXModulus1,
// src/NumbTh.cpp:1741-1804, two FFT multiplications with a precomputed inverse) uses
// Phi_m * Psi = X^m - 1 with Phi_m palindromic, so rev(Phi_m)^-1 = -Psi mod X^m: the quotient
// is rev(top(x) * (-Psi)) and the remainder x - Q*Phi_m, two more exact convolutions.
// Included by engine.hip only.
#pragma once
#include "conv_dev.h"

namespace hx {

// cbuf[(ri*batch+b)][Nc] <- x_i * powers[i] (i < phim), 0 elsewhere.  in: poly rows [row][b][phim]
__global__ void __launch_bounds__(256)
blue_pre_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ cbuf, NttRows rows, PtrList bp,
                int batch, uint32_t phim, uint32_t nc, int inverse_tables)
{
  const unsigned ri = blockIdx.y / (unsigned)batch, b = blockIdx.y % (unsigned)batch;
  const BluePrimeDev* P = (const BluePrimeDev*)bp.p[ri];
  const uint64_t q = P->q;
  const TW* pw = inverse_tables ? P->ipowers : P->powers;
  const uint64_t* src = in + ((size_t)rows.row[ri] * batch + b) * phim;
  uint64_t* dst = cbuf + ((size_t)ri * batch + b) * nc;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x)
    dst[i] = i < phim ? shoup_full(src[i], pw[i], q) : 0;
}

// inverse direction: place y_j at index t_j of a length-m vector (src/CModulus.cpp:559-563)
__global__ void __launch_bounds__(256)
blue_scatter_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ cbuf, NttRows rows,
                    PtrList bp, int batch, uint32_t phim, uint32_t m, uint32_t nc,
                    const int32_t* __restrict__ zidx)
{
  const unsigned ri = blockIdx.y / (unsigned)batch, b = blockIdx.y % (unsigned)batch;
  const BluePrimeDev* P = (const BluePrimeDev*)bp.p[ri];
  const uint64_t q = P->q;
  const uint64_t* src = in + ((size_t)rows.row[ri] * batch + b) * phim;
  uint64_t* dst = cbuf + ((size_t)ri * batch + b) * nc;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nc; i += gridDim.x * blockDim.x) {
    uint64_t v = 0;
    if (i < m) {
      int32_t j = zidx[i];
      if (j >= 0)
        v = shoup_full(src[j], P->ipowers[i], q);
    }
    dst[i] = v;
  }
}

// window / fold + second twist.  gather != 0: forward transform, out = poly rows [row][b][phim]
// holding X_k for k = zms[j]; gather == 0: inverse, out = xfull[(ri*batch+b)][mpad], all k < m.
__global__ void __launch_bounds__(256)
blue_post_kernel(const uint64_t* __restrict__ cbuf, uint64_t* __restrict__ out, NttRows rows, PtrList bp,
                 int batch, uint32_t phim, uint32_t m, uint32_t nc, uint32_t mpad,
                 const uint32_t* __restrict__ zms, int gather)
{
  const unsigned ri = blockIdx.y / (unsigned)batch, b = blockIdx.y % (unsigned)batch;
  const BluePrimeDev* P = (const BluePrimeDev*)bp.p[ri];
  const uint64_t q = P->q;
  const TW* pw = gather ? P->powers : P->ipowers;
  const uint64_t* h = cbuf + ((size_t)ri * batch + b) * nc;
  uint64_t* dst = gather ? out + ((size_t)rows.row[ri] * batch + b) * phim
                         : out + ((size_t)ri * batch + b) * mpad;
  const uint32_t n = gather ? phim : m;
  const bool odd = (m & 1u) != 0;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const uint32_t i = gather ? zms[j] : j;
    // odd m : coefficients 0..2(m-1), folded mod x^m - 1  (src/bluestein.cpp:166-187)
    // even m: coefficients m-1 .. 2(m-1)                   (src/bluestein.cpp:189-199)
    uint64_t v = odd ? addm(h[i], h[i + m], q) : h[m - 1 + i];
    dst[j] = shoup_full(v, pw[i], q);
  }
}

// The same for a radix-4 split result still in its four sub-blocks (the fused convolution kernel's
// output, qbuf[((ri*4+g)*batch+b)][Q]): value i of the 4Q-point result is rebuilt on the fly from the
// sub-block outputs at position i mod Q (split_inv4_one), so the inverse split pass, the window / fold
// and the second twist are one kernel.
__global__ void __launch_bounds__(256)
blue_post4_kernel(const uint64_t* __restrict__ qbuf, uint64_t* __restrict__ out, NttRows rows, PtrList bp, PtrList cps,
                  int batch, uint32_t phim, uint32_t m, uint32_t logq, uint32_t mpad,
                  const uint32_t* __restrict__ zms, int gather)
{
  const unsigned ri = blockIdx.y / (unsigned)batch, b = blockIdx.y % (unsigned)batch;
  const BluePrimeDev* P = (const BluePrimeDev*)bp.p[ri];
  const ConvPrimeDev* C = (const ConvPrimeDev*)cps.p[ri];
  const uint64_t q = P->q;
  const SplitTW S = C->S;
  const TW* pw = gather ? P->powers : P->ipowers;
  const uint32_t Q = 1u << logq;
  const uint64_t* sub[4];
  for (int g = 0; g < 4; g++)
    sub[g] = qbuf + (((size_t)ri * 4 + g) * batch + b) * Q;
  uint64_t* dst = gather ? out + ((size_t)rows.row[ri] * batch + b) * phim
                         : out + ((size_t)ri * batch + b) * mpad;
  const uint32_t n = gather ? phim : m;
  const bool odd = (m & 1u) != 0;
  auto value = [&](uint32_t i) {
    const uint32_t p = i & (Q - 1u);
    const uint64_t c[4] = {sub[0][p], sub[1][p], sub[2][p], sub[3][p]};
    return split_inv4_one(c, S, q, i >> logq);
  };
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const uint32_t i = gather ? zms[j] : j;
    const uint64_t v = odd ? addm(value(i), value(i + m), q) : value(m - 1 + i);
    dst[j] = shoup_full(v, pw[i], q);
  }
}

// data[(u*batch+b)][n] *= hat_u[n]   (u = conv unit: a row, or a (row, quarter) pair)
__global__ void __launch_bounds__(256)
conv_pointwise_kernel(uint64_t* __restrict__ data, PtrList hats, PtrList cps, int units_per_row,
                      int batch, uint32_t n)
{
  const unsigned u = blockIdx.y / (unsigned)batch, b = blockIdx.y % (unsigned)batch;
  const ConvPrimeDev* C = (const ConvPrimeDev*)cps.p[u / units_per_row];
  const uint64_t q = C->q, mu = C->mu;
  const uint32_t k = C->k;
  const uint64_t* hat = (const uint64_t*)hats.p[u / units_per_row] + (size_t)(u % units_per_row) * n;
  uint64_t* d = data + ((size_t)u * batch + b) * n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    d[i] = mul_mod(d[i], hat[i], q, mu, k);
}

// radix-4 split (conv_core.h): cbuf[(ri*batch+b)][4Q]  <->  qbuf[((ri*4+g)*batch+b)][Q]
__global__ void __launch_bounds__(256)
conv_split_kernel(uint64_t* __restrict__ cbuf, uint64_t* __restrict__ qbuf, PtrList cps, int batch,
                  uint32_t Q, int inverse)
{
  const unsigned ri = blockIdx.y / (unsigned)batch, b = blockIdx.y % (unsigned)batch;
  const ConvPrimeDev* C = (const ConvPrimeDev*)cps.p[ri];
  const uint64_t q = C->q;
  const SplitTW S = C->S;
  uint64_t* c = cbuf + ((size_t)ri * batch + b) * 4 * Q;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < Q; p += gridDim.x * blockDim.x) {
    uint64_t o[4];
    if (!inverse) {
      split_fwd4(c[p], c[p + Q], c[p + 2 * Q], c[p + 3 * Q], S, q, o);
      for (int g = 0; g < 4; g++)
        qbuf[(((size_t)ri * 4 + g) * batch + b) * Q + p] = o[g];
    } else {
      uint64_t in4[4];
      for (int g = 0; g < 4; g++)
        in4[g] = qbuf[(((size_t)ri * 4 + g) * batch + b) * Q + p];
      split_inv4(in4, S, q, o);
      c[p] = o[0];
      c[p + Q] = o[1];
      c[p + 2 * Q] = o[2];
      c[p + 3 * Q] = o[3];
    }
  }
}

// radix-8 / radix-16 split: cbuf[(ri*batch+b)][R Q]  <->  qbuf[((ri*R+g)*batch+b)][Q]
template <int LS>
__device__ __forceinline__ const SplitTWN<LS>& split_tw(const ConvPrimeDev* C)
{
  if constexpr (LS == 3)
    return C->S8;
  else
    return C->S16;
}
template <int LS>
__global__ void __launch_bounds__(256)
conv_splitN_kernel(uint64_t* __restrict__ cbuf, uint64_t* __restrict__ qbuf, PtrList cps, int batch,
                   uint32_t Q, int inverse)
{
  constexpr int R = 1 << LS;
  const unsigned ri = blockIdx.y / (unsigned)batch, b = blockIdx.y % (unsigned)batch;
  const ConvPrimeDev* C = (const ConvPrimeDev*)cps.p[ri];
  const uint64_t q = C->q;
  uint64_t* c = cbuf + ((size_t)ri * batch + b) * R * Q;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < Q; p += gridDim.x * blockDim.x) {
    uint64_t e[R];
    if (!inverse) {
#pragma unroll
      for (int g = 0; g < R; g++)
        e[g] = c[p + (size_t)g * Q];
      split_fwdN<LS>(e, split_tw<LS>(C), q);
#pragma unroll
      for (int g = 0; g < R; g++)
        qbuf[(((size_t)ri * R + g) * batch + b) * Q + p] = e[g];
    } else {
#pragma unroll
      for (int g = 0; g < R; g++)
        e[g] = qbuf[(((size_t)ri * R + g) * batch + b) * Q + p];
      split_invN<LS>(e, split_tw<LS>(C), q);
#pragma unroll
      for (int g = 0; g < R; g++)
        c[p + (size_t)g * Q] = e[g];
    }
  }
}

// ---- power-of-two rings beyond one row kernel (N = 2^16 .. 2^19, m up to 2^20): Cmodulus::FFT /
// iFFT (src/CModulus.cpp:389-426, 493-553; the reference only needs k <= NTL's MaxRoot, :108-110) as
// S = 4 / 8 / 16 sub-transforms of Q = N/S points (conv_core.h) with NATURAL order on both sides:
//   forward   big_pre  : in[row][b][p + gQ] --first log2(S) Cooley-Tukey stages--> qbuf[((ri*S+g)*batch+b)][p]
//             row kernels on the S sub-blocks (each natural order within its block)
//             big_post : out[row][b][j*S + brev(g)] = qbuf[g][j]       (the bit reversal of the top index bits)
//   inverse   big_pre  : qbuf[g][j] = in[row][b][j*S + brev(g)]; sub-inverses (scaled by 1/Q);
//             big_post : last log2(S) Gentleman-Sande stages incl. 1/S -> out[row][b][p + gQ]
template <int S>
__global__ void __launch_bounds__(256)
big_pre_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ qbuf, NttRows rows, PtrList cps, int batch,
               uint32_t Q, int inverse)
{
  constexpr int LS = S == 16 ? 4 : (S == 8 ? 3 : 2);
  const unsigned ri = blockIdx.y / (unsigned)batch, b = blockIdx.y % (unsigned)batch;
  const ConvPrimeDev* C = (const ConvPrimeDev*)cps.p[ri];
  const uint64_t q = C->q;
  const uint64_t* src = in + ((size_t)rows.row[ri] * batch + b) * S * Q;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < Q; p += gridDim.x * blockDim.x) {
    uint64_t e[S];
    if (!inverse) {
#pragma unroll
      for (int g = 0; g < S; g++)
        e[g] = src[p + (size_t)g * Q];
      if constexpr (S >= 8) {
        split_fwdN<LS>(e, split_tw<LS>(C), q);
      } else {
        uint64_t o[4];
        split_fwd4(e[0], e[1], e[2], e[3], C->S, q, o);
#pragma unroll
        for (int g = 0; g < 4; g++)
          e[g] = o[g];
      }
    } else {
#pragma unroll
      for (int g = 0; g < S; g++)
        e[g] = src[(size_t)p * S + brev_bits((unsigned)g, LS)];
    }
#pragma unroll
    for (int g = 0; g < S; g++)
      qbuf[(((size_t)ri * S + g) * batch + b) * Q + p] = e[g];
  }
}
template <int S>
__global__ void __launch_bounds__(256)
big_post_kernel(const uint64_t* __restrict__ qbuf, uint64_t* __restrict__ out, NttRows rows, PtrList cps, int batch,
                uint32_t Q, int inverse)
{
  constexpr int LS = S == 16 ? 4 : (S == 8 ? 3 : 2);
  const unsigned ri = blockIdx.y / (unsigned)batch, b = blockIdx.y % (unsigned)batch;
  const ConvPrimeDev* C = (const ConvPrimeDev*)cps.p[ri];
  const uint64_t q = C->q;
  uint64_t* dst = out + ((size_t)rows.row[ri] * batch + b) * S * Q;
  for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < Q; p += gridDim.x * blockDim.x) {
    uint64_t e[S];
#pragma unroll
    for (int g = 0; g < S; g++)
      e[g] = qbuf[(((size_t)ri * S + g) * batch + b) * Q + p];
    if (!inverse) {
#pragma unroll
      for (int g = 0; g < S; g++)
        dst[(size_t)p * S + brev_bits((unsigned)g, LS)] = e[g];
    } else {
      if constexpr (S >= 8) {
        split_invN<LS>(e, split_tw<LS>(C), q);
      } else {
        const uint64_t c4[4] = {e[0], e[1], e[2], e[3]};
        uint64_t a[4];
        split_inv4(c4, C->S, q, a);
#pragma unroll
        for (int g = 0; g < 4; g++)
          e[g] = a[g];
      }
#pragma unroll
      for (int g = 0; g < S; g++)
        dst[p + (size_t)g * Q] = e[g];
    }
  }
}

// ---- convolution modulo a prime WITHOUT the 2-power roots (q-1 not divisible by 2^(bk+1)):
// the exact integer convolution is taken modulo three auxiliary NTT primes A0, A1, A2 (each
// 2^20 | A-1, A > 2^59) and recombined -- what NTL does for a zz_p modulus that is not an FFT prime.
// Inputs are < q, at most 2^17 terms: every coefficient is < 2^17 q^2 < 2^137 < A0 A1 A2.
struct Crt3Dev {
  uint64_t A[3], muA[3];   // aux primes, floor(2^(2k)/A) (k = 60)
  uint64_t inv01;          // A0^-1 mod A1
  uint64_t a0m2, inv012;   // A0 mod A2, (A0 A1)^-1 mod A2
  uint64_t q, mu, mu64;    // the real modulus
  uint32_t k;
  uint64_t a0q, a01q;      // A0 mod q, A0 A1 mod q
};
// dst[i] = src[i] mod A (src < 2^60 < 2A)
__global__ void __launch_bounds__(256) aux_load_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst,
                                                       size_t n, uint64_t A)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint64_t x = src[i];
    dst[i] = x >= A ? x - A : x;
  }
}
// out[i] = CRT(t0[i] mod A0, t1[i] mod A1, t2[i] mod A2) mod q  (Garner mixed radix, exact)
__global__ void __launch_bounds__(256) crt3_kernel(const uint64_t* __restrict__ t0, const uint64_t* __restrict__ t1,
                                                   const uint64_t* __restrict__ t2, uint64_t* __restrict__ out,
                                                   size_t n, Crt3Dev C)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t d0 = t0[i];
    const uint64_t A1 = C.A[1], A2 = C.A[2];
    const uint64_t d0m1 = d0 >= A1 ? d0 - A1 : d0;       // aux primes are within 2x of each other
    const uint64_t d1 = mul_mod(sub_mod(t1[i], d0m1, A1), C.inv01, A1, C.muA[1], 60);
    const uint64_t d0m2 = d0 >= A2 ? d0 - A2 : d0, d1m2 = d1 >= A2 ? d1 - A2 : d1;
    uint64_t v = sub_mod(t2[i], d0m2, A2);
    v = sub_mod(v, mul_mod(C.a0m2, d1m2, A2, C.muA[2], 60), A2);
    const uint64_t d2 = mul_mod(v, C.inv012, A2, C.muA[2], 60);
    // value = d0 + A0 d1 + A0 A1 d2, reduced modulo q
    uint64_t r = red64(d0, C.q, C.mu64);
    r = add_mod(r, mul_mod(C.a0q, red64(d1, C.q, C.mu64), C.q, C.mu, C.k), C.q);
    r = add_mod(r, mul_mod(C.a01q, red64(d2, C.q, C.mu64), C.q, C.mu, C.k), C.q);
    out[i] = r;
  }
}

// dst[(ri*batch+b)][nd] <- k <= d ? src[(ri*batch+b)][base -/+ k] : 0   (reversal + zero padding)
//   mode 0: top of x reversed : dst[k] = x[m-1-k]        (src stride mpad, base = m-1)
//   mode 1: Q from rev_d(Q)   : dst[k] = s[d-k]          (src stride ns,   base = d)
__global__ void __launch_bounds__(256)
blue_rev_kernel(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, int batch, uint32_t sstride,
                uint32_t base, uint32_t d, uint32_t nd)
{
  const size_t seg = blockIdx.y;
  const uint64_t* s = src + seg * sstride;
  uint64_t* o = dst + seg * nd;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < nd; k += gridDim.x * blockDim.x)
    o[k] = k <= d ? s[base - k] : 0;
}

// out[row][b][i] = (xfull[i] - (Q*Phi)[i]) * m^-1,  i < phim     (rem Phi_m, then x *= mm_inv)
__global__ void __launch_bounds__(256)
blue_final_kernel(const uint64_t* __restrict__ xfull, const uint64_t* __restrict__ qphi,
                  uint64_t* __restrict__ out, NttRows rows, PtrList bp, int batch, uint32_t phim,
                  uint32_t mpad, uint32_t n2)
{
  const unsigned ri = blockIdx.y / (unsigned)batch, b = blockIdx.y % (unsigned)batch;
  const BluePrimeDev* P = (const BluePrimeDev*)bp.p[ri];
  const uint64_t q = P->q;
  const TW minv = P->minv;
  const uint64_t* x = xfull + ((size_t)ri * batch + b) * mpad;
  const uint64_t* z = qphi + ((size_t)ri * batch + b) * n2;
  uint64_t* dst = out + ((size_t)rows.row[ri] * batch + b) * phim;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < phim; i += gridDim.x * blockDim.x)
    dst[i] = shoup_full(subm(x[i], z[i], q), minv, q);
}

}  // namespace hx

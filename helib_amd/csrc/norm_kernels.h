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
// wanted, so the output order is irrelevant: decimation-in-frequency radix-2, in place, no
// bit-reversal.  An N-point transform is split by log2(S) DIF levels applied while loading into S
// independent H-point transforms (H = N/S <= 8192 complex doubles = 128 KiB of LDS), one workgroup
// each; every workgroup folds its own maximum into out2[row] with an atomic max on the bit pattern
// (non-negative doubles order like unsigned integers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hx {

constexpr int NORM_MAX_LOGH = 13;  // 2^13 complex doubles = 128 KiB LDS
constexpr int NORM_THREADS = 1024;

// f: rows x N doubles; wtab[k] = W^k for k < N (W^(k+N) = -W^k); out2[row] must be zeroed.
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
  // H-point DIF with root U = W^(2S): stage of half-length len uses U^(j*H/(2 len)) = W^(j*N/len)
  for (unsigned len = H >> 1, sh = logh - 1; len >= 1; len >>= 1, sh--) {
    const unsigned tstride = N >> sh;  // N/len
    for (unsigned bf = tid; bf < (H >> 1); bf += nth) {
      const unsigned j = bf & (len - 1), k = ((bf >> sh) << (sh + 1)) + j;
      const double ar = re[k], ai = im[k], br = re[k + len], bi = im[k + len];
      const double2 w = wtab[j * tstride];
      const double dr = ar - br, di = ai - bi;
      re[k] = ar + br;
      im[k] = ai + bi;
      re[k + len] = dr * w.x - di * w.y;
      im[k + len] = dr * w.y + di * w.x;
    }
    __syncthreads();
    if (len == 1)
      break;
  }
  double mx = 0;
  for (unsigned i = tid; i < H; i += nth) {
    const double n2 = re[i] * re[i] + im[i] * im[i];
    mx = n2 > mx ? n2 : mx;
  }
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
    atomicMax(out2 + row, (unsigned long long)__double_as_longlong(mx));
  }
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

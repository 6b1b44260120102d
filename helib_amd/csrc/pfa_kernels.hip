// pfa_kernels.hip -- the Good-Thomas x Rader row kernels for m = 21845 = 5 * 17 * 257 on gfx950 (pfa_core.h has the
// method, the phase functions and the table builders): Cmodulus::FFT / iFFT of the general-m branch
// (src/CModulus.cpp:431-443, 555-577) as ONE launch per direction instead of the Bluestein chirp convolution
// (src/bluestein.cpp:134-201).  One 1024-thread workgroup per (row, batch element); 140 KB of LDS, so one workgroup
// per CU at four waves per SIMD.  Callers: engine.hip (bluestein_rows).
#include "dev_common.h"
#include "pfa_dev.h"
#include "ntt_kernel_util.h"
#include "prof.h"

namespace hx {

// MODE 0: forward.  1: inverse up to X (m words per row; rem Phi_m by the convolution kernels).  2: inverse with
// rem Phi_m and 1/m fused behind it (binomial passes in the LDS), the default.
constexpr int pfa_nphases(int mode) { return mode == 0 ? pfa::FWD_PHASES : (mode == 1 ? pfa::INV_PHASES : pfa::INV_REM_PHASES); }
constexpr int pfa_lds_words(int mode)
{
  return mode == 0 ? pfa::LDS_WORDS
                   : (mode == 1 ? pfa::INV_LDS_WORDS : (pfa::LDS_WORDS > pfa::REM_LDS_WORDS ? pfa::LDS_WORDS : pfa::REM_LDS_WORDS));
}
template <int MODE, int PH, class Q>
__device__ __forceinline__ void pfa_phases(unsigned tid, pfa::St& s, uint64_t* lds, const pfa::Args& A, const Q& q)
{
  if constexpr (PH < pfa_nphases(MODE)) {
    if constexpr (MODE == 2)
      pfa::inv_rem<PH>(tid, s, lds, A, q);
    else if constexpr (MODE == 1)
      pfa::inv<PH>(tid, s, lds, A, q);
    else
      pfa::fwd<PH>(tid, s, lds, A, q);
    if constexpr (PH + 1 < pfa_nphases(MODE))
      __syncthreads();
    pfa_phases<MODE, PH + 1>(tid, s, lds, A, q);
  }
}

// PROTH: every row of the launch is on a Proth-form prime (the word-wise Montgomery product) / on any other prime (the
// generic one).  One arithmetic per kernel: with both behind a uniform branch in one kernel the Proth rows ran 30 %
// slower (94 -> 130 us per 512 rows: twice the code, four more registers).
template <int MODE, bool PROTH>
__global__ void __launch_bounds__(pfa::NT)
pfa_row_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, PfaRows R, const PrimeDev* __restrict__ primes,
               const uint16_t* __restrict__ pos2, const uint16_t* __restrict__ dlog3, const uint16_t* __restrict__ gpow3,
               unsigned batch, unsigned out_stride)
{
  extern __shared__ __attribute__((aligned(16))) uint64_t pfa_lds[];
  const unsigned wid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned b = wid % batch, ri = wid / batch;
  const PrimeDev* pd = primes + uniform_u16(R.prime, ri);
  pfa::Args A;
  A.tab = R.tab[ri];
  A.pos2 = pos2;
  A.dlog3 = dlog3;
  A.gpow3 = gpow3;
  const size_t polyseg = ((size_t)uniform_u16(R.row, ri) * batch + b) * (size_t)pfa::PHI;
  A.src = in + polyseg;
  A.dst = MODE == 1 ? out + ((size_t)ri * batch + b) * (size_t)out_stride : out + polyseg;
  pfa::St s;
  if constexpr (PROTH) {
    const pfa::QCP q = pfa::make_qcp(pd->q, pd->mu64);
    pfa_phases<MODE, 0>(threadIdx.x, s, pfa_lds, A, q);
  } else {
    const pfa::QCG q = pfa::make_qcg(pd->q, pd->mu64);
    pfa_phases<MODE, 0>(threadIdx.x, s, pfa_lds, A, q);
  }
}

template <int MODE, bool PROTH>
static hipError_t launch_pfa(const uint64_t* in, uint64_t* out, const PfaRows& R, int nrows, const PrimeDev* primes,
                             const uint16_t* pos2, const uint16_t* dlog3, const uint16_t* gpow3, int batch,
                             unsigned out_stride, hipStream_t st)
{
  constexpr size_t lds_bytes = (size_t)pfa_lds_words(MODE) * 8;
  hipError_t e = hxp::dyn_lds((const void*)pfa_row_kernel<MODE, PROTH>, (int)lds_bytes);
  if (e != hipSuccess)
    return e;
  HX_LAUNCH((pfa_row_kernel<MODE, PROTH>), dim3((unsigned)nrows * (unsigned)batch), dim3(pfa::NT), lds_bytes, st, in, out, R, primes,
            pos2, dlog3, gpow3, (unsigned)batch, out_stride);
  return hipGetLastError();
}
// mode 0 / 2: out = poly rows; mode 1: out = X[(ri * batch + b)][out_stride], m words each
// proth: the rows of R are all on Proth-form primes (PrimeDev::proth) / all on other primes
hipError_t launch_pfa_rows(int mode, bool proth, const uint64_t* in, uint64_t* out, const PfaRows& R, int nrows,
                           const PrimeDev* primes, const uint16_t* pos2, const uint16_t* dlog3, const uint16_t* gpow3,
                           int batch, unsigned out_stride, hipStream_t st)
{
  switch (mode * 2 + (proth ? 1 : 0)) {
    case 0: return launch_pfa<0, false>(in, out, R, nrows, primes, pos2, dlog3, gpow3, batch, out_stride, st);
    case 1: return launch_pfa<0, true>(in, out, R, nrows, primes, pos2, dlog3, gpow3, batch, out_stride, st);
    case 2: return launch_pfa<1, false>(in, out, R, nrows, primes, pos2, dlog3, gpow3, batch, out_stride, st);
    case 3: return launch_pfa<1, true>(in, out, R, nrows, primes, pos2, dlog3, gpow3, batch, out_stride, st);
    case 4: return launch_pfa<2, false>(in, out, R, nrows, primes, pos2, dlog3, gpow3, batch, out_stride, st);
    case 5: return launch_pfa<2, true>(in, out, R, nrows, primes, pos2, dlog3, gpow3, batch, out_stride, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace hx

// pfa_dev.h -- launch descriptor of the Good-Thomas x Rader row kernels (pfa_kernels.hip), shared with engine.hip.
#pragma once
#include "dev_common.h"
#include "pfa_core.h"

namespace hx {

constexpr int PFA_MAXROWS = 64;
struct PfaRows {                        // per launch: up to PFA_MAXROWS rows
  const uint64_t* tab[PFA_MAXROWS];     // the row's prime table (pfa::TAB_WORDS words, pfa::host::build_prime_table)
  uint16_t row[PFA_MAXROWS];            // row of the poly buffer
  uint16_t prime[PFA_MAXROWS];          // index into hx_ctx::d_primes
};

// mode 0: forward; 1: inverse up to X (rem Phi_m elsewhere); 2: inverse with rem Phi_m and 1/m fused
// proth: all rows on Proth-form primes (PrimeDev::proth), or none (one arithmetic per launch)
hipError_t launch_pfa_rows(int mode, bool proth, const uint64_t* in, uint64_t* out, const PfaRows& R, int nrows,
                           const PrimeDev* primes, const uint16_t* pos2, const uint16_t* dlog3, const uint16_t* gpow3,
                           int batch, unsigned out_stride, hipStream_t st);

}  // namespace hx

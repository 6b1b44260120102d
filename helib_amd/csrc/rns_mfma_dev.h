// rns_mfma_dev.h -- launch of the matrix-core basis extension (rns_mfma_kernels.hip), shared with engine.hip.
#pragma once
#include "dev_common.h"
#include "rns_types.h"

namespace hx {

// same contract as rns_extend_wide_kernel (rns_kernels.h): P.mfma_steps != 0, A.redo set; the caller runs the Garner
// pass over the redo list behind it
hipError_t launch_rns_extend_mfma(const ExtPlanDev& P, const ExtArgs& A, size_t row_words, hipStream_t st);

}  // namespace hx

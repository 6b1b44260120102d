// helib_amd_intel.hpp -- `namespace intel` with the eight signatures of HElib's HEXL shim
// (src/intelExt.h:20-59), inline over the C ABI (hx_intel_* in include/helib_amd.h).
//
// A HElib translation unit built with USE_INTEL_HEXL calls these by name (src/CModulus.cpp:385, 514;
// src/DoubleCRT.cpp:144-195, 329): with this header on the include path in place of src/intelExt.h --
// and libhelib_amd.so on the link line in place of HEXL -- those call sites compile and link
// unchanged.  Semantics are the shim's: host pointers, synchronous, void-returning (an engine error
// becomes a std::runtime_error, where HEXL would abort), in place allowed, and FFTFwd / FFTRev1 are HEXL's
// transforms in HEXL's order under HEXL's root: FFTFwd(n, q) carries no root, the NTT
// object picks MinimalPrimitiveRoot(2n, q); its output is in BIT-REVERSED evaluation order,
// out[i] = f(psi^(2*brev(i)+1)), and FFTRev1 consumes that order -- which is why the reference runs
// BitReverseCopy after the forward call (src/CModulus.cpp:421-426) and before the inverse one (:510).
// With those two copies in place, as they are in the reference, rows are the natural ones that
// DoubleCRT::automorph and the wire format index; they differ from an NTL build's rows by the root
// only.  One PCIe round trip per call: this is the link-compatibility layer, not the fast path (that is
// include/helib_amd.hpp, whole DoubleCRT objects resident on the device).
#ifndef HELIB_AMD_INTEL_HPP
#define HELIB_AMD_INTEL_HPP
#include <stdexcept>
#include <string>

#include "helib_amd.h"

namespace intel {
namespace detail {
inline void check(int rc)
{
  if (rc != HX_OK)
    throw std::runtime_error(std::string("intel:: shim over helib_amd: ") + hx_last_error());
}
}  // namespace detail

inline void FFTFwd(long* output, const long* input, long n, long q) { detail::check(hx_intel_FFTFwd(output, input, n, q)); }
inline void FFTRev1(long* output, const long* input, long n, long q) { detail::check(hx_intel_FFTRev1(output, input, n, q)); }

inline void EltwiseAddMod(long* result, const long* operand1, const long* operand2, long n, long modulus)
{
  detail::check(hx_intel_EltwiseAddMod(result, operand1, operand2, n, modulus));
}
inline void EltwiseAddMod(long* result, const long* operand, long scalar, long n, long modulus)
{
  detail::check(hx_intel_EltwiseAddModScalar(result, operand, scalar, n, modulus));
}
inline void EltwiseSubMod(long* result, const long* operand1, const long* operand2, long n, long modulus)
{
  detail::check(hx_intel_EltwiseSubMod(result, operand1, operand2, n, modulus));
}
inline void EltwiseSubMod(long* result, const long* operand, long scalar, long n, long modulus)
{
  detail::check(hx_intel_EltwiseSubModScalar(result, operand, scalar, n, modulus));
}
inline void EltwiseMultMod(long* result, const long* operand1, const long* operand2, long n, long modulus)
{
  detail::check(hx_intel_EltwiseMultMod(result, operand1, operand2, n, modulus));
}
inline void EltwiseMultMod(long* result, const long* operand, long scalar, long n, long modulus)
{
  detail::check(hx_intel_EltwiseMultModScalar(result, operand, scalar, n, modulus));
}

}  // namespace intel
#endif  // HELIB_AMD_INTEL_HPP

// ntt_dispatch.hip -- the row-transform entry points of ntt_kernels.hip, picking the translation unit of the ring
// size: ntt_kernels.hip is compiled three times (-DHX_NTT_ONLY=13|14|15, in parallel), each unit exporting its
// entry points with the suffix _L13 / _L14 / _L15 (see the top of that file).  Rings below 2^13 (the small-ring
// kernel, one workgroup per row in LDS) are in every unit; the 2^14 one serves them.
#include "dev_common.h"

namespace hx {

#define HX_FOR_SIZES(X) X(13) X(14) X(15)

// name, parameter list (with logn first), argument list
#define HX_ENTRIES(E, S)                                                                                                   \
  E(S, launch_ntt_pow2,                                                                                                    \
    (int logn, bool inverse, const uint64_t* in, uint64_t* out, const NttRows& rows, int nrows, int batch,               \
     const PrimeDev* primes, const TW* tw_arena, hipStream_t st),                                                          \
    (logn, inverse, in, out, rows, nrows, batch, primes, tw_arena, st))                                                    \
  E(S, launch_ntt_pow2_lazy_in,                                                                                            \
    (int logn, const uint64_t* in, uint64_t* out, const NttRows& rows, int nrows, int batch, const PrimeDev* primes,      \
     const TW* tw_arena, hipStream_t st),                                                                                  \
    (logn, in, out, rows, nrows, batch, primes, tw_arena, st))                                                             \
  E(S, launch_moddown_pow2,                                                                                                \
    (int logn, const PolyBases& data, const PolyBases& out, int drop_row, int drop_prime, const NttRows& keep, int nkeep, \
     int batch, const ModDownPrep& P, const ModDownApply& A, const PrimeDev* primes, const TW* tw_arena, hipStream_t st), \
    (logn, data, out, drop_row, drop_prime, keep, nkeep, batch, P, A, primes, tw_arena, st))                               \
  E(S, launch_moddown_prep_pow2,                                                                                           \
    (int logn, const PolyBases& polys, int drop_row, int drop_prime, int batch, const ModDownPrep& P,                     \
     const PrimeDev* primes, const TW* tw_arena, hipStream_t st),                                                          \
    (logn, polys, drop_row, drop_prime, batch, P, primes, tw_arena, st))                                                   \
  E(S, launch_moddown_prep_multi_pow2,                                                                                     \
    (int logn, const PolyBases& polys, const PrepMulti& M, int ndrop, int batch, const ModDownPrep& P,                    \
     const PrimeDev* primes, const TW* tw_arena, hipStream_t st),                                                          \
    (logn, polys, M, ndrop, batch, P, primes, tw_arena, st))                                                               \
  E(S, launch_moddown_apply_plain_pow2,                                                                                    \
    (int logn, const PolyBases& polys, const PolyBases& outs, const NttRows& keep, int nkeep, int batch,                  \
     const ModDownApply& A, const PrimeDev* primes, const TW* tw_arena, hipStream_t st),                                   \
    (logn, polys, outs, keep, nkeep, batch, A, primes, tw_arena, st))                                                      \
  E(S, launch_moddown_prep_multi_tensor_pow2,                                                                              \
    (int logn, const TensorSrc& T, const PrepMulti& M, int ndrop, int batch, const ModDownPrep& P,                        \
     const PrimeDev* primes, const TW* tw_arena, hipStream_t st),                                                          \
    (logn, T, M, ndrop, batch, P, primes, tw_arena, st))                                                                   \
  E(S, launch_moddown_apply_plain_tensor_pow2,                                                                             \
    (int logn, const TensorSrc& T, const PolyBases& outs, const NttRows& keep, int nkeep, int batch,                      \
     const ModDownApply& A, const PrimeDev* primes, const TW* tw_arena, hipStream_t st),                                   \
    (logn, T, outs, keep, nkeep, batch, A, primes, tw_arena, st))                                                          \
  E(S, launch_moddown_tensor_pow2,                                                                                         \
    (int logn, const TensorSrc& T, const PolyBases& outs, int drop_row, int drop_prime, const NttRows& keep, int nkeep,   \
     int batch, const ModDownPrep& P, const ModDownApply& A, const PrimeDev* primes, const TW* tw_arena, hipStream_t st), \
    (logn, T, outs, drop_row, drop_prime, keep, nkeep, batch, P, A, primes, tw_arena, st))                                 \
  E(S, launch_ntt_inv_mul_pow2,                                                                                            \
    (int logn, const uint64_t* a, const uint64_t* b, uint64_t* out, const NttRows& rows, int nrows, int batch,            \
     const PrimeDev* primes, const TW* tw_arena, hipStream_t st),                                                          \
    (logn, a, b, out, rows, nrows, batch, primes, tw_arena, st))

// declarations of the per-size units' entry points
#define HX_DECL(S, name, params, args) hipError_t name##_L##S params;
#define HX_DECL_ALL(S) HX_ENTRIES(HX_DECL, S)
HX_FOR_SIZES(HX_DECL_ALL)

// the public entry points
#define HX_DEF(S, name, params, args)               \
  hipError_t name params                            \
  {                                                 \
    switch (logn) {                                 \
      case 13: return name##_L13 args;              \
      case 15: return name##_L15 args;              \
      default: return name##_L14 args;              \
    }                                               \
  }
HX_ENTRIES(HX_DEF, 0)

}  // namespace hx

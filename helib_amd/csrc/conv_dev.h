// conv_dev.h -- device-side descriptors of the Bluestein convolutions shared by engine.hip
// (bluestein.h: element-wise kernels) and ntt_kernels.hip (the fused convolution row kernel).
#pragma once
#include "dev_common.h"
#include "conv_core.h"

namespace hx {

// per (prime, conv size) constants
struct ConvPrimeDev {
  uint64_t q, mu, mu64;
  uint32_t k, logn;
  SplitTW S;           // only for split sizes (radix 4: 2^16, 2^17)
  SplitTW8 S8;         // radix 8: 2^18
  SplitTW16 S16;       // radix 16: 2^19
};
// per prime constants of the Bluestein transform
struct BluePrimeDev {
  uint64_t q;
  const TW* powers;    // [m] root^(i^2)      (src/bluestein.cpp:94-98)
  const TW* ipowers;   // [m] rInv^(i^2)
  TW minv;             // m^-1 mod q          (src/CModulus.cpp:574-577)
};
struct PtrList {
  const void* p[MAX_ROWS];
};


// ---- the fused convolution kernel (ntt_kernels.hip: ntt_conv_kernel) ----
// One workgroup = one 2^LOGN-point sub-transform of one (row, batch element): element source -> forward
// transform -> product with the precomputed transform of the fixed operand (Shoup pairs) -> inverse
// transform -> store, without leaving the register file in between.  It replaces, for the Bluestein
// convolutions, the chain  pre/scatter/reverse pass -> split pass -> forward launches -> pointwise pass ->
// inverse launches  of round 1/2 (seven global passes per row for a forward transform).
constexpr int CONV_MAXROWS = 16;
struct ConvRows {                        // per launch: up to CONV_MAXROWS rows, `split` units each
  const BluePrimeDev* bp[CONV_MAXROWS];
  const TW* hat[CONV_MAXROWS];           // transform of the fixed operand as {w, floor(w 2^64 / q)}, [g][Q]
  const ConvPrimeDev* cp[CONV_MAXROWS];
  uint16_t row[CONV_MAXROWS];            // row of the poly buffer (sources / destinations in poly layout)
  uint16_t pd[CONV_MAXROWS * 4];         // twiddle-table entry of unit (row, g)
};
enum ConvSrc : uint32_t {
  CONV_SRC_BLUE_PRE = 0,   // x_i * powers[i], i < phim                       (forward chirp input, bluestein.cpp:153-156)
  CONV_SRC_SCATTER = 1,    // y_j * ipowers[t_j] at index t_j of Z_m^*, else 0  (CModulus.cpp:559-563)
  CONV_SRC_REV = 2,        // i <= d ? s[base - i] : 0                        (the reversals of rem Phi_m)
};
enum ConvDst : uint32_t {
  CONV_DST_SUB = 0,        // out[(unit * batch + b) * Q + p]                 (sub-block buffer / plain row)
  CONV_DST_FINAL = 1,      // poly row: (aux[p] - v) * m^-1 for p < phim       (rem Phi_m + the 1/m of CModulus.cpp:574-577)
};
struct ConvRowArgs {
  const uint64_t* in;
  uint64_t* out;
  const uint64_t* aux;
  const int32_t* zidx;
  uint32_t src_mode, dst_mode, split, batch;
  uint32_t phim, m, in_stride, aux_stride;   // strides: words between (row, batch) segments of in / aux (non-poly layouts)
  uint32_t d, base;
  uint32_t alias;   // CONV_DST_FINAL: the product was taken modulo X^Q + 1 with Q < m -- coefficient i also carries
                    // -(Q Phi_m)_(i + Q) = -x_(i + Q), which the store adds back (engine.hip, hx_ctx::n3)
};

}  // namespace hx

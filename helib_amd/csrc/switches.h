// switches.h -- every environment switch of the library, in ONE place (host code only).
//
// None of them is needed in production.  They select the older or more generic form of a path so that tests can
// reach it and same-box A/B measurements can compare it (tools/gpu_calls.sh ab:...); every one leaves results
// bit-identical.  They are listed for users in include/helib_amd.h and read ONCE PER CONTEXT: hx_ctx_create()
// snapshots the environment INTO THE CONTEXT (hx_ctx::sw = hxs::read()), every later decision of the library reads
// that context's copy -- so a test process can change a switch between two contexts without touching the first (no
// process-global state: round 4 kept one struct that every hx_ctx_create rewrote under the running threads of earlier
// contexts), and nothing consults the environment on a hot path.
// (The C++ host has one switch of its own, HX_NO_LAZY_TENSOR in include/helib_amd_ctxt.hpp, read once per process.)
#pragma once
#include <cstdlib>

namespace hxs {

struct Switches {
  // exact RNS kernels (rns_kernels.h)
  bool no_hps = false;           // HX_NO_HPS=1          fast kernels (<= 16 sources): Garner instead of the HPS front end
  double hps_eps = 1.0 / (double)(1u << 30);   // HX_HPS_EPS=x   distance from 0, 1/2, 1 below which an HPS quotient is redone
  int hps_min_n = 9;             // HX_HPS_MIN_N=n       fast kernels: HPS form from n source primes on
  int brk_hps_min_n = 5;         // HX_BRK_HPS_MIN_N=n   ... the digit kernel's own threshold (round 6: digits of 5 - 8 primes run faster on
                                 //                      the HPS front end + its redo launch than on Garner; HX_HPS_MIN_N overrides both)
  bool no_lazy_rns = false;      // HX_NO_LAZY_RNS=1     no 128-bit lazy sums / one-subtraction Garner steps
  bool no_fast_break = false;    // HX_NO_FAST_BREAK=1   generic break_digits_kernel instead of the fast one
  bool no_fast_extend = false;   // HX_NO_FAST_EXTEND=1  generic rns_extend_kernel instead of rns_extend_fast_kernel
  bool no_proth_rns = false;     // HX_NO_PROTH_RNS=1    fast kernels: Barrett / Shoup products on Proth-form primes too (HX_NO_PROTH implies it)
  bool no_wide_extend = false;   // HX_NO_WIDE_EXTEND=1  generic rns_extend_kernel<40> instead of rns_extend_wide_kernel (17..40 sources)
  bool no_mfma_ext = false;      // HX_NO_MFMA_EXT=1     rns_extend_wide_kernel (VALU limb products) instead of the matrix-core form
                                 //                      rns_extend_mfma_kernel (17..40 sources; HX_NO_WIDE_EXTEND implies it)
  int mfma_min_n = 9;            // HX_MFMA_MIN_N=n      ... from n source primes on (9 .. 16: the fast kernels' plans; 17 .. 40 always)
  int brk_lds_pad_rows = 0;      // HX_BRK_LDS_PAD=n     digit kernel: n more (unused) LDS rows per thread -- lowers its occupancy, an A/B probe
  // fused ciphertext-level paths
  bool no_tensor_multi = false;  // HX_NO_TENSOR_MULTI=1 tensor product + several-primes mod-switch as two steps
  bool no_mulrelin_fuse = false; // HX_NO_MULRELIN_FUSE=1 hx_mul_relin with a tensor pass
  // row transforms (ntt_core.h)
  bool half15 = false;           // HX_HALF15=1          N = 2^15 forward rows (out of place) as two 2^14-point workgroups per row: measured 3-4 % SLOWER
                                 //                      than the one-workgroup kernel (profiles/r06_ab_half_row_forward_2p15.json); kept as a probe
  bool no_proth = false;         // HX_NO_PROTH=1        Shoup butterflies on every row (Proth-form primes included)
  // general m
  bool blue_old = false;         // HX_BLUE_OLD=1        Bluestein as the round-2 chain of passes instead of ntt_conv_kernel
  bool no_pfa = false;           // HX_NO_PFA=1          m = 21845: Bluestein instead of the Good-Thomas x Rader kernels (pfa_core.h)
  bool pfa_no_rem = false;       // HX_PFA_NO_REM=1      ... their inverse stops at X: rem Phi_m on the convolution kernels, not fused
  // canonical-embedding norm kernels
  bool norm_async = false;       // HX_NORM_ASYNC=1      norm kernels on a side stream
  bool norm_old = false;         // HX_NORM_OLD=1        N = 2^14 / 2^15: the LDS-pass kernels instead of the radix-16 ones
  bool norm_plain = false;       // HX_NORM_PLAIN=1      no split into sub-transforms above 2^14 points
  bool norm_memcpy = false;      // HX_NORM_MEMCPY=1     norm read-back by hipMemcpy instead of mapped host memory
  // host waits
  int wait_poll_us = 2000;       // HX_WAIT_POLL_US=n    how long a norm read-back is polled for before the thread sleeps in hipEventSynchronize
  // diagnostics
  bool arena_trace = false;      // HX_ARENA_TRACE=1     one line on stderr per hipMalloc the slab arena makes
};

// The environment as it is now, by value: hx_ctx_create() stores it in the context (hx_ctx::sw) and every later decision
// made for that context reads its own copy -- a context created later, under a changed environment, does not touch it.
inline Switches read()
{
  Switches s;
  auto on = [](const char* name) { return std::getenv(name) != nullptr; };
  s.no_hps = on("HX_NO_HPS");
  if (const char* e = std::getenv("HX_HPS_EPS"))
    s.hps_eps = std::atof(e);
  if (const char* e = std::getenv("HX_HPS_MIN_N"))
    s.hps_min_n = s.brk_hps_min_n = std::atoi(e);
  if (const char* e = std::getenv("HX_BRK_HPS_MIN_N"))
    s.brk_hps_min_n = std::atoi(e);
  s.no_lazy_rns = on("HX_NO_LAZY_RNS");
  s.no_fast_break = on("HX_NO_FAST_BREAK");
  s.no_fast_extend = on("HX_NO_FAST_EXTEND");
  s.no_proth_rns = on("HX_NO_PROTH_RNS");
  s.no_wide_extend = on("HX_NO_WIDE_EXTEND");
  s.no_mfma_ext = on("HX_NO_MFMA_EXT") || s.no_wide_extend;
  if (const char* e = std::getenv("HX_MFMA_MIN_N"))
    s.mfma_min_n = std::atoi(e);
  if (const char* e = std::getenv("HX_BRK_LDS_PAD"))
    s.brk_lds_pad_rows = std::atoi(e);
  s.half15 = on("HX_HALF15");
  s.no_tensor_multi = on("HX_NO_TENSOR_MULTI");
  s.no_mulrelin_fuse = on("HX_NO_MULRELIN_FUSE");
  s.no_proth = on("HX_NO_PROTH");
  s.blue_old = on("HX_BLUE_OLD");
  s.no_pfa = on("HX_NO_PFA") || s.blue_old;
  s.pfa_no_rem = on("HX_PFA_NO_REM");
  s.norm_async = on("HX_NORM_ASYNC");
  s.norm_old = on("HX_NORM_OLD");
  s.norm_plain = on("HX_NORM_PLAIN");
  s.norm_memcpy = on("HX_NORM_MEMCPY");
  if (const char* e = std::getenv("HX_WAIT_POLL_US"))
    s.wait_poll_us = std::atoi(e);
  s.arena_trace = on("HX_ARENA_TRACE");
  return s;
}

}  // namespace hxs

// rns_types.h -- plan and launch descriptors of the exact RNS basis-extension kernels, shared by rns_kernels.h
// (compiled into engine.hip) and rns_mfma_kernels.hip (the matrix-core form, a unit of its own).
#pragma once
#include "dev_common.h"

namespace hx {

// words of one target's record: 8 header words + the multipliers padded to a multiple of four (host and device agree)
__host__ __device__ inline int wide_stride(int n) { return 8 + ((n + 3) & ~3); }

struct ExtPlanDev {
  int n;                   // source primes
  int nt;                  // target primes
  ro_u64 src_q;   // [n]
  ro_u64 src_mu64;  // [n]
  ro_tw ginv;          // [n*n]  ginv[k*n+l] = p_l^-1 mod p_k   (l<k)
  ro_u64 half;    // [n] mixed-radix digits of (P-1)/2
  ro_u64 tgt_q;   // [nt]
  ro_u64 tgt_mu64;  // [nt]
  ro_u64 tgt_mu;  // [nt]  Barrett mu (128-bit products)
  ro_u32 tgt_k;   // [nt]
  ro_u64 pmod;    // [nt]  P mod t
  ro_tw W;             // [nt*n] W[t*n+k] = (p_0..p_{k-1}) mod t
  ro_tw upd;           // [nt]  P^-1 mod t (Shoup) for the breakIntoDigits update
  // BGV mod-switch correction (scaleDownToSet): ptxt = 0 disables
  uint64_t ptxt, ptxt_mu64, pinv_ptxt /* P^-1 mod ptxt */, pmod_ptxt /* P mod ptxt */;
  uint64_t ptxt_mu;  uint32_t ptxt_k;
  ro_tw Wp;            // [n] (p_0..p_{k-1}) mod ptxt
  // strength reductions decided by the host from the actual primes:
  uint32_t garner_cs;        // every source residue is < 2*p_k for every later source prime p_k:
                             // "a_l mod p_k" is one conditional subtraction
  ro_u32 tgt_lazy;  // [nt] 1: sum_k q_k <= 8*q_t, so the target residue can be taken from
                             // the 128-bit sum of the n products with ONE Barrett reduction
  ro_f64 src_rq;      // [n] 1.0 / q_k (host-rounded), for the value/P fraction
  ro_u64 tgt_mu63;  // [nt] floor(2^(63+k) / q_t), k = bitlen(q_t)  (red128_q8)
  uint32_t fast_ok;          // break_digits_fast_kernel's preconditions hold for this plan:
                             // garner_cs, every prime > 2^32 (32-bit reciprocals), n <= 8
  uint32_t fast16_ok;        // the same with n <= 16: rns_extend_fast_kernel
  uint32_t corr_unit;        // scaled plan (tables carry P^-1, so "P mod t" is 1) and ptxtSpace <= every
                             // target prime: the plaintext-space correction of a target is the balanced
                             // remainder itself -- no reduction, no 128-bit Barrett product per target
  ro_u32 tgt_chunk7;  // [nt] 1: every source prime <= q_t, so the limb sum may be taken 7
                               // terms at a time with the previous remainder carried (r + 7 p q < 8 q^2)
  // HPS front end of the fast kernels (hps_ok): instead of Garner's n(n-1)/2 dependent products,
  //   y_k = a_k (P/p_k)^-1 mod p_k,  value = sum_k y_k (P/p_k) - v P,  v = floor(sum_k y_k / p_k),
  // the quotient v taken from a double-precision sum that is trusted only when its fractional part
  // is at least hps_eps away from 0, 1/2 and 1 (centring compares with 1/2); a wavefront with an
  // untrusted lane takes the Garner path, so the result is exact either way.
  ro_tw hps_inv;       // [n] (P/p_k)^-1 mod p_k
  ro_u64 tgt_pack_hps; // [nt][10 + 2n] as tgt_pack with the multipliers (P/p_k) mod t (scaled: / P)
  ro_tw Wp_hps;        // [n] (P/p_k) mod ptxt
  double hps_eps;
  uint32_t hps_ok;
  ro_u64 wide_pack;   // [nt][wide_stride(n)] rns_extend_wide_kernel's record of one target: 8 header words, HPS multipliers as limb pairs
  uint32_t wide_ok;   // 16 < n <= 40 sources, every prime in (2^32, 2^60), the HPS tables exist
  // the same extension on the matrix cores (rns_mfma_kernels.hip; layout and builder: mfma_ext.h): the multipliers as
  // balanced 8-bit limbs in V_MFMA_I32_32X32X32_I8 operand order, the accumulators' start values; mfma_steps = 0: not built
  const void* mfma_a;        // [tiles][mfma_steps x 64 operand vectors + 16 vectors of per-target constants] (mfma_ext.h)
  uint32_t mfma_steps;       // K = 32 steps of one tile: ceil((n + 1) / 4)
  // Proth-form primes (q = qh 2^32 + 1, ntt_core.h is_proth32; round 5): the fast kernels' products as Montgomery
  // products -- a target whose record carries TgtRec::mont() holds its multipliers, -P mod t and P^-1 mod t times 2^64,
  // and its limb sum is reduced by mont_redc128 (two multiply-adds for the Barrett's seven multiplications, or the ten
  // of red128_any); src_mont: every SOURCE prime has the form too and the Garner steps run on ginv_m
  ro_u64 ginv_m;      // [n*n] ginv_m[k*n+l] = p_l^-1 2^64 mod p_k   (l<k)
  uint32_t src_mont;
  // the readers of LAZY output words take rows of Proth-form primes at the tight bounds the Montgomery target sums
  // deliver (2q digits, 4q several-primes mod-down: ntt_kernels.hip BufIOT / ModDownIO): true only when every
  // Proth-form target of this plan carries TgtRec::mont and every source prime is below 2^60 (host-derived; a plan
  // without it gets canonical output -- ADVICE r5)
  uint32_t lazy_tight_ok;
  ro_u64 tgt_pack;    // [nt][10 + 2n] everything the fast kernels need of one target in ONE record
                      // (TgtRec): the loop over targets then makes one scalar-memory round trip per
                      // target instead of one per table (q, P mod t, flags, k, mu, W row: six
                      // dependent s_load / s_waitcnt pairs per iteration before)
};

constexpr int EXT_MAXSRC = 192;  // source primes of one extension (a whole bits=6400 chain: 143)
struct ExtArgs {
  const uint64_t* src;       // coefficient rows, [row][batch][N]
  uint64_t* dst;             // output rows,      [row][batch][N]
  uint64_t* upd;             // rows updated in place (breakIntoDigits), may alias src
  uint16_t src_row[EXT_MAXSRC];      // row of source prime k inside src
  uint16_t own_dst_row[EXT_MAXSRC];  // where to copy the source residue in dst (0xffff: no copy)
  uint16_t dst_row[MAX_ROWS];  // output row of target t inside dst
  uint16_t upd_row[MAX_ROWS];  // row inside upd to update (0xffff: none)
  int nu;                    // targets [0,nu) are the ones with an upd_row (host orders them first)
  double* frac;              // optional [batch][N]: (centred, corrected) value / P as a double
                             // (the fdelta of src/Ctxt.cpp:466-478 for scaleDownToSet)
  uint32_t* redo;            // fast kernels: [0] = count, [1..] = coefficient indices (see ExtRep below); or null
  uint32_t lazy_out;         // 1: the dst words may stay unreduced in [0,8q) -- their only reader is a forward row
                             // transform declared with LOAD_BOUND 8 (the several-primes mod-down's apply kernels);
                             // honoured by rns_extend_fast_kernel, never together with upd rows
};

}  // namespace hx

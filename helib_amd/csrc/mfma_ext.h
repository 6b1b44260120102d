// mfma_ext.h -- the exact basis extension from many source primes as an int8 matrix product on the matrix cores
// (rns_mfma_kernels.hip): limb split, table layout and recombination, as pure functions shared by the kernel, the host
// table builder (engine.hip ext_plan_get) and the CPU restatement (tests/cpp/mfma_ext_test.cpp).
//
// What is computed (src/DoubleCRT.cpp:565-599 addPrimes, :1464-1516 scaleDownToSet, :479-561 breakIntoDigits at the
// reference's own benchmark chain, benchmarks/bgv_basic.cpp:247): for every target prime t
//     r_t = ( sum_k y_k W_kt  -  cnt P )  mod t,      y_k < p_k < 2^60 (the HPS form's digits), W_kt = (P/p_k) mod t,
// n = 17..40 sources, up to 143 targets: the one dense-GEMM-shaped step of this path.  rns_extend_wide_kernel does it
// with four 30-bit-limb multiply-adds per (source, target) term -- 144 per (coefficient, target) at n = 36.
//
// The matrix form.  y_k = sum_a y_ka 2^(8a) in BALANCED 8-bit limbs (y_ka in [-128,127], the top one in [0,16]); for
// every limb position the multiplier is reduced first,
//     W'_(k,a),t = W_kt 2^(8a) mod t  =  sum_b w_(k,a),t,b 2^(8b)      (balanced limbs again),
// so that  sum_k y_k W_kt  ==  sum_b 2^(8b) S_tb  (mod t),   S_tb = sum_(k,a) y_ka w_(k,a),t,b:
// ONE integer matrix product  S[(t,b)][coefficient] = Wl[(t,b)][(k,a)] x Yl[(k,a)][coefficient]  with
// |S_tb| <= K 2^14 (K = 8 source slots x limbs = 32 per MFMA step), exact in the i32 accumulators of
// V_MFMA_I32_32X32X32_I8.  -cnt P enters as one more source slot (y = cnt < 128, multiplier -P mod t), and the
// accumulators start from base + delta_tb with  sum_b (base + delta_tb) 2^(8b) == 0 (mod t),  base = K 2^14, so every
// S'_tb = S_tb + base + delta_tb is a non-negative 24-bit number and the recombined value is congruent to the sum.
//
// Layout of one MFMA step (32 rows x 32 columns x K = 32):
//   * columns = 32 coefficients; lanes l and l + 32 hold the same column (h = l >> 5 selects the K half)
//   * byte i of lane (col, h) of step j  <->  source slot 4 j + 2 h + (i >> 3), limb i & 7      (operand B: y)
//   * row r of the A operand <-> target  4 tile + 2 ((r >> 2) & 1) + (r >> 4),  limb 4 ((r >> 3) & 1) + (r & 3):
//     the C/D layout (col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h) then leaves lane (col, h) with ALL
//     eight limb sums of targets 4 tile + 2 h + s in registers 8 s .. 8 s + 7 (s = 0, 1) -- no cross-lane traffic
//     between the matrix product and the reduction modulo t.
#pragma once
#include <cstdint>
#include <vector>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define HX_MFX_HD __host__ __device__
#else
#define HX_MFX_HD
#endif

namespace hx {
namespace mfx {

constexpr int TILE_TARGETS = 4;      // targets per 32-row tile (8 limb rows each)
constexpr int MIN_STEPS = 5, MAX_STEPS = 11;   // n = 17..40 sources (+ the cnt slot) in groups of four

HX_MFX_HD inline int steps_for(int n) { return (n + 1 + 3) / 4; }   // K = 32 steps; slot 4 steps - 1 carries cnt
HX_MFX_HD inline int tiles_for(int nt) { return (nt + TILE_TARGETS - 1) / TILE_TARGETS; }
HX_MFX_HD inline uint32_t acc_base(int steps) { return (uint32_t)steps * 32u * 16384u; }   // >= max |S_tb|

// balanced 8-bit limbs of y < 2^63, packed: byte a = y_ka as a signed byte (the top byte takes no offset, it is small
// and non-negative).  y + C carries the +128 offsets through the bytes, ^ C turns each offset digit into the signed one.
HX_MFX_HD inline uint64_t pack_balanced(uint64_t y)
{
  const uint64_t C = 0x0080808080808080ull;
  return (y + C) ^ C;
}
HX_MFX_HD inline int limb_of(uint64_t packed, int a) { return (int)(int8_t)(uint8_t)(packed >> (8 * a)); }

HX_MFX_HD inline int row_target(int r) { return 2 * ((r >> 2) & 1) + (r >> 4); }
HX_MFX_HD inline int row_limb(int r) { return 4 * ((r >> 3) & 1) + (r & 3); }
HX_MFX_HD inline int row_of(int target_in_tile, int limb)
{
  return 16 * (target_in_tile & 1) + 4 * (target_in_tile >> 1) + 8 * (limb >> 2) + (limb & 3);
}
// accumulator register `reg` (0..15) of lane half h holds row:
HX_MFX_HD inline int cd_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// sizes of the two device tables
inline size_t a_table_bytes(int nt, int steps) { return (size_t)tiles_for(nt) * (size_t)steps * 64 * 16; }
inline size_t init_table_words(int nt) { return (size_t)tiles_for(nt) * 2 * 16; }

// Host: the tables of one plan.
//   tq[t]     target primes (2^32 < t < 2^60)
//   w[t*n+k]  the multipliers (P/p_k) mod t (a scaled plan: / P), < t
//   negp[t]   -P mod t (a scaled plan: -1 mod t), the multiplier of the cnt slot
//   a_out     [tiles][steps][64 lanes][16 bytes]   signed bytes
//   init_out  [tiles][2 lane halves][16 registers] u32
inline void build_tables(int n, int nt, const uint64_t* tq, const uint64_t* w, const uint64_t* negp,
                         std::vector<int8_t>& a_out, std::vector<uint32_t>& init_out)
{
  typedef unsigned __int128 u128;
  const int steps = steps_for(n), slots = 4 * steps, tiles = tiles_for(nt);
  a_out.assign(a_table_bytes(nt, steps), 0);
  init_out.assign(init_table_words(nt), 0);
  const uint32_t base = acc_base(steps);
  for (int tau = 0; tau < tiles; tau++) {
    for (int tt = 0; tt < TILE_TARGETS; tt++) {
      const int t = tau * TILE_TARGETS + tt;
      if (t >= nt)
        continue;
      const uint64_t q = tq[t];
      for (int k = 0; k < slots; k++) {
        uint64_t m = 0;
        if (k < n)
          m = w[(size_t)t * n + k];
        else if (k == slots - 1)
          m = negp[t];
        if (m == 0)
          continue;
        const int j = k >> 2, h = (k >> 1) & 1;
        for (int a = 0; a < 8; a++) {
          const uint64_t wa = (uint64_t)(((u128)m << (8 * a)) % q);
          const uint64_t packed = pack_balanced(wa);
          for (int b = 0; b < 8; b++) {
            const int lane = row_of(tt, b) + 32 * h, byte = 8 * (k & 1) + a;
            a_out[(((size_t)tau * steps + j) * 64 + lane) * 16 + byte] = (int8_t)limb_of(packed, b);
          }
        }
      }
      // accumulator start: base + the bytes of D = -(base sum_b 2^(8b)) mod t
      u128 vb = 0;
      for (int b = 0; b < 8; b++)
        vb += (u128)base << (8 * b);
      const uint64_t D = (uint64_t)((q - (uint64_t)(vb % q)) % q);
      for (int b = 0; b < 8; b++) {
        const uint32_t delta = b < 7 ? (uint32_t)((D >> (8 * b)) & 0xffu) : (uint32_t)(D >> 56);
        const int r = row_of(tt, b), h = (r >> 2) & 1, reg = (r & 3) + 4 * (r >> 3);
        init_out[((size_t)tau * 2 + h) * 16 + reg] = base + delta;
      }
    }
  }
}

// Recombination: the eight start-offset limb sums S'_b (each < 2^24) of one (coefficient, target) as the 80-bit
// value sum_b S'_b 2^(8b) = hi 2^64 + lo, hi < 2^16.  (Kernel and CPU restatement share it.)
struct V80 {
  uint64_t lo;
  uint32_t hi;
};
HX_MFX_HD inline V80 recombine(const uint32_t (&S)[8])
{
  const uint32_t a01 = S[0] + (S[1] << 8), a23 = S[2] + (S[3] << 8);   // < 2^32: S < 2^23.5
  const uint32_t a45 = S[4] + (S[5] << 8), a67 = S[6] + (S[7] << 8);
  const uint64_t L = (uint64_t)a23 * 65536u + a01, H = (uint64_t)a67 * 65536u + a45;   // < 2^48
  const uint64_t up = (L >> 32) + H;                                                   // < 2^48 + 2^16
  V80 v;
  v.lo = (uint64_t)(uint32_t)L | (up << 32);
  v.hi = (uint32_t)(up >> 32);
  return v;
}

}  // namespace mfx
}  // namespace hx

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
constexpr int MIN_STEPS = 2, MAX_STEPS = 11;   // n = 4..40 sources (+ the cnt slot) in groups of four

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

// The device table: per tile, steps x 64 operand vectors (16 bytes each: lane l of step j at vector j 64 + l) followed
// by EXTRA_VECS vectors of per-target constants -- the kernel stages a whole tile block through its LDS, so everything
// a tile needs arrives with its operands.  Lane half h owns vectors 8 h .. 8 h + 7 of the extras, as 32-bit words:
//   [0, 16)  the accumulator start values of its registers
//   16, 17   floor(2^80 / t) of its targets s = 0, 1 (0 when t < 2^48: that target takes the general reduction)
//   20 .. 23 t of s = 0 (low, high word), t of s = 1
//   24 .. 31 P^-1 mod t with its Shoup companion (the in-place update of breakIntoDigits), s = 0 then s = 1: w, floor(w 2^64 / t)
constexpr int EXTRA_VECS = 16;
constexpr int EX_INIT = 0, EX_MU80 = 16, EX_Q = 20, EX_UPD = 24;
HX_MFX_HD inline int tile_vecs(int steps) { return steps * 64 + EXTRA_VECS; }
inline size_t table_bytes(int nt, int steps) { return (size_t)tiles_for(nt) * (size_t)tile_vecs(steps) * 16; }
inline size_t a_byte_index(int steps, int tau, int j, int lane, int byte)
{
  return (((size_t)tau * tile_vecs(steps) + (size_t)j * 64 + lane) * 16) + byte;
}
inline size_t extra_word_index(int steps, int tau, int h, int w)   // in 32-bit words from the start of the table
{
  return ((size_t)tau * tile_vecs(steps) + (size_t)steps * 64 + 8 * h) * 4 + w;
}

// The 80-bit value V = hi 2^64 + lo (hi < 2^16) modulo t for 2^48 <= t < 2^60, congruent result in [0, 4t):
// x = floor(V / 2^48) < 2^32, mu80 = floor(2^80 / t) < 2^32, qh = floor(x mu80 / 2^32) is floor(V / t) or up to 3 less
// (V/t - x 2^48/t < 2^48/t <= 1; x (2^80/t - mu80) / 2^32 < 1; the floor 1), so lo - qh t (mod 2^64) is the value.
HX_MFX_HD inline uint64_t red80_lazy(uint64_t lo, uint32_t hi, uint64_t t, uint32_t mu80)
{
  const uint32_t x = (hi << 16) | (uint32_t)(lo >> 48);
  const uint32_t qh = (uint32_t)(((uint64_t)x * mu80) >> 32);
  return lo - (uint64_t)qh * t;
}

// Host: the table of one plan.
//   tq[t]     target primes (2^32 < t < 2^60)
//   w[t*n+k]  the multipliers (P/p_k) mod t (a scaled plan: / P), < t
//   negp[t]   -P mod t (a scaled plan: -1 mod t), the multiplier of the cnt slot
//   upd[2t], upd[2t+1]  P^-1 mod t and its Shoup companion
//   tab       table_bytes(nt, steps_for(n)) bytes, layout above
inline void build_tables(int n, int nt, const uint64_t* tq, const uint64_t* w, const uint64_t* negp, const uint64_t* upd,
                         std::vector<uint8_t>& tab)
{
  typedef unsigned __int128 u128;
  const int steps = steps_for(n), slots = 4 * steps, tiles = tiles_for(nt);
  tab.assign(table_bytes(nt, steps), 0);
  uint32_t* words = reinterpret_cast<uint32_t*>(tab.data());
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
          for (int b = 0; b < 8; b++)
            tab[a_byte_index(steps, tau, j, row_of(tt, b) + 32 * h, 8 * (k & 1) + a)] = (uint8_t)(int8_t)limb_of(packed, b);
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
        words[extra_word_index(steps, tau, h, EX_INIT + reg)] = base + delta;
      }
      const int h = tt >> 1, sidx = tt & 1;
      if (q >> 48)
        words[extra_word_index(steps, tau, h, EX_MU80 + sidx)] = (uint32_t)((((u128)1) << 80) / q);
      words[extra_word_index(steps, tau, h, EX_Q + 2 * sidx)] = (uint32_t)q;
      words[extra_word_index(steps, tau, h, EX_Q + 2 * sidx + 1)] = (uint32_t)(q >> 32);
      for (int e = 0; e < 2; e++) {
        words[extra_word_index(steps, tau, h, EX_UPD + 4 * sidx + 2 * e)] = (uint32_t)upd[2 * (size_t)t + e];
        words[extra_word_index(steps, tau, h, EX_UPD + 4 * sidx + 2 * e + 1)] = (uint32_t)(upd[2 * (size_t)t + e] >> 32);
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

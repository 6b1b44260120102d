// CPU restatement of the int8 matrix form of the many-source basis extension (helib_amd/csrc/mfma_ext.h;
// rns_mfma_kernels.hip is the kernel): the tables the host builds, V_MFMA_I32_32X32X32_I8 as a plain triple loop over
// its documented operand layout, the recombination and the reduction -- against sum_k y_k W_kt - cnt P mod t taken
// directly in 128-bit arithmetic.  TEST INFRASTRUCTURE: built by tests/, never linked into the product library.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../helib_amd/csrc/mfma_ext.h"

typedef unsigned __int128 u128;
using namespace hx::mfx;

static uint64_t rng_state = 1;
static uint64_t rnd()
{
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

// one step: D[row][col] = C[row][col] + sum over (h, byte) A[lane = row + 32 h][byte] * B[lane = col + 32 h][byte]
// (both operands use the same (lane half, byte) -> k map, so the contraction does not depend on what that map is)
static void mfma_step(const int8_t* A /*[64][16]*/, const int8_t* B /*[64][16]*/, int32_t (*acc)[32])
{
  for (int r = 0; r < 32; r++)
    for (int c = 0; c < 32; c++) {
      int32_t s = 0;
      for (int h = 0; h < 2; h++)
        for (int i = 0; i < 16; i++)
          s += (int32_t)A[(r + 32 * h) * 16 + i] * (int32_t)B[(c + 32 * h) * 16 + i];
      acc[r][c] += s;
    }
}

// returns 0 when every (coefficient, target) of a random instance agrees; otherwise a code that says what failed.
//   n sources of `sbits` bits, nt targets of tbits[t % 3] bits (odd numbers > 2^32 stand in for primes: nothing here
//   needs primality), 32 coefficients, cnt in [0, n + 1].  worst != 0: the y are all p_k - 1 / limbs at their extremes
extern "C" int mfma_ext_check(int n, int nt, int sbits, int tb0, int tb1, int tb2, uint64_t seed, int worst, uint32_t* max_acc)
{
  rng_state = seed * 0x9e3779b97f4a7c15ull + 1;
  const int tbits[3] = {tb0, tb1, tb2};
  const int steps = steps_for(n), slots = 4 * steps, tiles = tiles_for(nt);
  if (steps < MIN_STEPS || steps > MAX_STEPS)
    return 100;
  std::vector<uint64_t> p(n), tq(nt), w((size_t)nt * n), negp(nt), upd(2 * (size_t)nt);
  for (int k = 0; k < n; k++)
    p[k] = (((uint64_t)1 << (sbits - 1)) | (rnd() >> (65 - sbits)) | 1);
  for (int t = 0; t < nt; t++) {
    const int b = tbits[t % 3];
    tq[t] = (((uint64_t)1 << (b - 1)) | (rnd() >> (65 - b)) | 1);
    if (worst)
      tq[t] = (((uint64_t)1 << b) - 1) - 2 * (uint64_t)t;   // just below 2^b
    for (int k = 0; k < n; k++)
      w[(size_t)t * n + k] = worst == 2 ? tq[t] - 1 - (uint64_t)k : rnd() % tq[t];
    negp[t] = rnd() % tq[t];
    upd[2 * (size_t)t] = rnd() % tq[t];
    upd[2 * (size_t)t + 1] = rnd();
  }
  std::vector<uint8_t> tab;
  build_tables(n, nt, tq.data(), w.data(), negp.data(), upd.data(), tab);
  if (tab.size() != table_bytes(nt, steps) || tab.size() != (size_t)tiles * (steps * 64 + 16) * 16)
    return 101;
  const int8_t* A = reinterpret_cast<const int8_t*>(tab.data());
  const uint32_t* words = reinterpret_cast<const uint32_t*>(tab.data());
  // 32 coefficients
  std::vector<uint64_t> y((size_t)32 * slots, 0);
  std::vector<uint32_t> cnt(32);
  for (int c = 0; c < 32; c++) {
    for (int k = 0; k < n; k++)
      y[(size_t)c * slots + k] = worst ? p[k] - 1 - (uint64_t)(c & 1) : rnd() % p[k];
    cnt[c] = worst ? (uint32_t)(n + 1) : (uint32_t)(rnd() % (uint64_t)(n + 2));
    y[(size_t)c * slots + slots - 1] = cnt[c];
  }
  // operand B of every step: lane (col, h), byte i <-> slot 4 j + 2 h + (i >> 3), limb i & 7
  std::vector<int8_t> B((size_t)steps * 64 * 16);
  for (int j = 0; j < steps; j++)
    for (int lane = 0; lane < 64; lane++)
      for (int i = 0; i < 16; i++) {
        const int col = lane & 31, h = lane >> 5, k = 4 * j + 2 * h + (i >> 3), a = i & 7;
        B[((size_t)j * 64 + lane) * 16 + i] = (int8_t)limb_of(pack_balanced(y[(size_t)col * slots + k]), a);
      }
  // the packing is the balanced expansion: sum_a limb_a 2^(8a) == y
  for (int c = 0; c < 32; c++)
    for (int k = 0; k < slots; k++) {
      const uint64_t pk = pack_balanced(y[(size_t)c * slots + k]);
      __int128 v = 0;
      for (int a = 0; a < 8; a++)
        v += (__int128)limb_of(pk, a) << (8 * a);
      if (v != (__int128)y[(size_t)c * slots + k] || limb_of(pk, 7) < 0)
        return 102;
    }
  uint32_t amax = 0;
  for (int tau = 0; tau < tiles; tau++) {
    int32_t acc[32][32];
    for (int r = 0; r < 32; r++)
      for (int c = 0; c < 32; c++)
        acc[r][c] = (int32_t)words[extra_word_index(steps, tau, (r >> 2) & 1, EX_INIT + (r & 3) + 4 * (r >> 3))];
    for (int j = 0; j < steps; j++)
      mfma_step(&A[a_byte_index(steps, tau, j, 0, 0)], &B[(size_t)j * 64 * 16], acc);
    // lane (col, h): registers 8 s + b of its 16 = rows cd_row(8 s + b, h) = target 4 tau + 2 h + s, limb b
    for (int h = 0; h < 2; h++)
      for (int s = 0; s < 2; s++) {
        const int t = 4 * tau + 2 * h + s;
        if (t >= nt)
          continue;
        for (int c = 0; c < 32; c++) {
          uint32_t S[8];
          for (int b = 0; b < 8; b++) {
            const int row = cd_row(8 * s + b, h);
            if (row_target(row) != 2 * h + s || row_limb(row) != b)
              return 103;
            if (acc[row][c] < 0 || (uint32_t)acc[row][c] >= (2u * acc_base(steps) + 256u))
              return 104;
            S[b] = (uint32_t)acc[row][c];
            amax = S[b] > amax ? S[b] : amax;
          }
          const V80 v = recombine(S);
          // the recombination is the plain sum
          u128 plain = 0;
          for (int b = 0; b < 8; b++)
            plain += (u128)S[b] << (8 * b);
          if (plain != (((u128)v.hi << 64) | v.lo) || v.hi >= (1u << 16))
            return 105;
          const uint64_t q = tq[t];
          const uint64_t got = (uint64_t)(plain % q);
          u128 want = 0;
          for (int k = 0; k < n; k++)
            want = (want + (u128)y[(size_t)c * slots + k] * w[(size_t)t * n + k]) % q;
          want = (want + (u128)cnt[c] * negp[t]) % q;
          if (got != (uint64_t)want)
            return 106;
          // the short reduction of targets >= 2^48: congruent and below 4 t
          const uint32_t mu80 = words[extra_word_index(steps, tau, h, EX_MU80 + s)];
          const uint64_t qtab = words[extra_word_index(steps, tau, h, EX_Q + 2 * s)] |
                                ((uint64_t)words[extra_word_index(steps, tau, h, EX_Q + 2 * s + 1)] << 32);
          if (qtab != q)
            return 109;
          for (int e = 0; e < 2; e++)
            if ((words[extra_word_index(steps, tau, h, EX_UPD + 4 * s + 2 * e)] |
                 ((uint64_t)words[extra_word_index(steps, tau, h, EX_UPD + 4 * s + 2 * e + 1)] << 32)) != upd[2 * (size_t)t + e])
              return 110;
          if ((mu80 != 0) != ((q >> 48) != 0))
            return 107;
          if (mu80) {
            const uint64_t lazy = red80_lazy(v.lo, v.hi, q, mu80);
            if (lazy >= 4 * q || lazy % q != got)
              return 108;
          }
        }
      }
  }
  if (max_acc)
    *max_acc = amax;
  return 0;
}

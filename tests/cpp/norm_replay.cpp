// norm_replay.cpp -- CPU replay of the register-tiled norm kernel's phases (helib_amd/csrc/norm_r16.h),
// thread by thread with the barriers where the kernel has them, against the definition
//   max_j | f(W^(2j+1)) |,  W = exp(2 pi i / 2N),  N = 16384
// evaluated directly (long double) at every point for a sparse polynomial and at sampled points for a
// dense one.  Checks the index maps of the three radix-16 passes, the twiddle strides, the padded LDS
// layout (one array, real parts then imaginary parts), the last stage as a lane exchange, and the pairing that
// forms each pair once through the exchange area.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../helib_amd/csrc/norm_r16.h"

using namespace hx;

// one transpose of embed_norm_r16_kernel: the real parts cross the ONE padded array, then the imaginary parts
// (write all, barrier, read all, barrier -- each loop over t below is one phase between two barriers)
template <class FROM, class TO>
static void transpose(std::vector<cplx16>& regs, std::vector<double>& sm, FROM from, TO to)
{
  const unsigned T = R16_THREADS;
  for (int comp = 0; comp < 2; comp++) {
    for (unsigned t = 0; t < T; t++)
      for (unsigned k = 0; k < 16; k++)
        sm[r16_pad(from(t, k))] = comp ? regs[(size_t)t * 16 + k].y : regs[(size_t)t * 16 + k].x;
    for (unsigned t = 0; t < T; t++)
      for (unsigned k = 0; k < 16; k++)
        (comp ? regs[(size_t)t * 16 + k].y : regs[(size_t)t * 16 + k].x) = sm[r16_pad(to(t, k))];
  }
}
static double replay(const std::vector<double>& f, const std::vector<tw16>& wtab)
{
  const unsigned T = R16_THREADS;
  std::vector<double> sm(R16_LDS_DOUBLES, 0.0);
  std::vector<cplx16> regs((size_t)T * 16);
  // load + pre-twist + pass A
  for (unsigned t = 0; t < T; t++) {
    cplx16 v[16];
    for (unsigned k = 0; k < 16; k++) {
      const unsigned i = r16_pos_A(t, k);
      const double a = f[2 * i], b = f[2 * i + 1];
      const tw16 w = k == 0 ? wtab[2 * t] : r16_cmul(wtab[2 * t], wtab[1024u * k]);   // W^(2i) = W^(2t) W^(1024 k)
      v[k].x = a * w.x - b * w.y;
      v[k].y = a * w.y + b * w.x;
    }
    r16_pass<9>(v, t, wtab.data());
    for (unsigned k = 0; k < 16; k++)
      regs[(size_t)t * 16 + k] = v[k];
  }
  transpose(regs, sm, r16_pos_A, r16_pos_B);
  for (unsigned t = 0; t < T; t++) {
    cplx16 v[16];
    for (unsigned k = 0; k < 16; k++)
      v[k] = regs[(size_t)t * 16 + k];
    r16_pass<5>(v, t & 31u, wtab.data());
    for (unsigned k = 0; k < 16; k++)
      regs[(size_t)t * 16 + k] = v[k];
  }
  transpose(regs, sm, r16_pos_B, r16_pos_C);
  for (unsigned t = 0; t < T; t++) {
    cplx16 v[16];
    for (unsigned k = 0; k < 16; k++)
      v[k] = regs[(size_t)t * 16 + k];
    r16_pass<1>(v, t & 1u, wtab.data());
    for (unsigned k = 0; k < 16; k++)
      regs[(size_t)t * 16 + k] = v[k];
  }
  // last stage: the lane exchange with t ^ 1 (all lanes read the partner's value of before the exchange)
  {
    std::vector<cplx16> z(regs.size());
    for (unsigned t = 0; t < T; t++)
      for (unsigned k = 0; k < 16; k++)
        z[(size_t)t * 16 + k] = r16_last_lane(regs[(size_t)t * 16 + k], regs[(size_t)(t ^ 1u) * 16 + k], t);
    regs.swap(z);
  }
  // barrier; the upper halves into the array; barrier; pairing
  for (unsigned t = 0; t < T; t++)
    for (unsigned kk = 0; kk < 8; kk++) {
      sm[r16_xchg_idx(t, kk)] = regs[(size_t)t * 16 + 8 + kk].x;
      sm[R16_XCHG_IM + r16_xchg_idx(t, kk)] = regs[(size_t)t * 16 + 8 + kk].y;
    }
  double mx = 0;
  for (unsigned t = 0; t < T; t++) {
    const tw16 wth = wtab[r16_pair_tw_thread(t)];
    for (unsigned k = 0; k < 8; k++) {
      const unsigned o = r16_xchg_idx(T - 1u - t, 7u - k);
      const cplx16 partner{sm[o], sm[R16_XCHG_IM + o]};
      const tw16 w = k == 0 ? wth : r16_cmul(wth, wtab[r16_pair_tw_k(k)]);
      const double n2 = r16_pair_norm2(regs[(size_t)t * 16 + k], partner, w);
      mx = n2 > mx ? n2 : mx;
    }
  }
  return std::sqrt(mx);
}

// N = 2^15, both sub-transforms at once (embed_norm_r16x2_kernel): half h of the 1024 threads = sub-transform h
static double replay_x2(const std::vector<double>& f, const std::vector<tw16>& wtab)
{
  const unsigned T = R16_THREADS;
  std::vector<double> sm[2] = {std::vector<double>(R16_LDS_DOUBLES, 0.0), std::vector<double>(R16_LDS_DOUBLES, 0.0)};
  std::vector<cplx16> regs[2] = {std::vector<cplx16>((size_t)T * 16), std::vector<cplx16>((size_t)T * 16)};
  for (unsigned h = 0; h < 2; h++) {
    for (unsigned t = 0; t < T; t++) {
      cplx16 v[16];
      for (unsigned k = 0; k < 16; k++)
        v[k] = r16_split_input(f.data(), wtab.data(), r16_pos_A(t, k), h);
      r16_pass<9, 15>(v, t, wtab.data());
      for (unsigned k = 0; k < 16; k++)
        regs[h][(size_t)t * 16 + k] = v[k];
    }
    transpose(regs[h], sm[h], r16_pos_A, r16_pos_B);
    for (unsigned t = 0; t < T; t++) {
      cplx16 v[16];
      for (unsigned k = 0; k < 16; k++)
        v[k] = regs[h][(size_t)t * 16 + k];
      r16_pass<5, 15>(v, t & 31u, wtab.data());
      for (unsigned k = 0; k < 16; k++)
        regs[h][(size_t)t * 16 + k] = v[k];
    }
    transpose(regs[h], sm[h], r16_pos_B, r16_pos_C);
    for (unsigned t = 0; t < T; t++) {
      cplx16 v[16];
      for (unsigned k = 0; k < 16; k++)
        v[k] = regs[h][(size_t)t * 16 + k];
      r16_pass<1, 15>(v, t & 1u, wtab.data());
      for (unsigned k = 0; k < 16; k++)
        regs[h][(size_t)t * 16 + k] = v[k];
    }
    std::vector<cplx16> z(regs[h].size());
    for (unsigned t = 0; t < T; t++)
      for (unsigned k = 0; k < 16; k++)
        z[(size_t)t * 16 + k] = r16_last_lane(regs[h][(size_t)t * 16 + k], regs[h][(size_t)(t ^ 1u) * 16 + k], t);
    regs[h].swap(z);
    for (unsigned t = 0; t < T; t++)
      for (unsigned kk = 0; kk < 8; kk++) {
        sm[h][r16_xchg_idx(t, kk)] = regs[h][(size_t)t * 16 + 8 + kk].x;
        sm[h][R16_XCHG_IM + r16_xchg_idx(t, kk)] = regs[h][(size_t)t * 16 + 8 + kk].y;
      }
  }
  double mx = 0;
  for (unsigned h = 0; h < 2; h++)
    for (unsigned t = 0; t < T; t++) {
      const tw16 wth = wtab[r16x2_pair_tw_thread(h, t)];
      for (unsigned k = 0; k < 8; k++) {
        const unsigned o = r16_xchg_idx(T - 1u - t, 7u - k);
        const cplx16 other{sm[1 - h][o], sm[1 - h][R16_XCHG_IM + o]};
        const double n2 = r16x2_pair(h, regs[h][(size_t)t * 16 + k], other, wth, wtab[r16x2_pair_tw_k(k)], k);
        mx = n2 > mx ? n2 : mx;
      }
    }
  return std::sqrt(mx);
}

static std::vector<long double> g_cos, g_sin;   // W^e, e < 2N
static long double direct_at(const std::vector<double>& f, unsigned j)
{
  const unsigned N = R16_N;
  long double sr = 0, si = 0;
  for (unsigned i = 0; i < N; i++) {
    if (f[i] == 0.0)
      continue;
    const unsigned long e = ((unsigned long)i * (2ul * j + 1ul)) % (2ul * N);
    sr += f[i] * g_cos[e];
    si += f[i] * g_sin[e];
  }
  return sqrtl(sr * sr + si * si);
}

int main()
{
  const unsigned N = R16_N;
  const long double two_pi = 6.283185307179586476925286766559005768394L;
  std::vector<tw16> wtab(N);
  for (unsigned k = 0; k < N; k++) {
    const long double ang = two_pi * (long double)k / (long double)(2 * N);
    wtab[k] = {(double)cosl(ang), (double)sinl(ang)};
  }
  {
    const long double pi = 3.141592653589793238462643383279502884L;
    g_cos.resize(2 * N);
    g_sin.resize(2 * N);
    for (unsigned e = 0; e < 2 * N; e++) {
      g_cos[e] = cosl(pi * (long double)e / (long double)N);
      g_sin[e] = sinl(pi * (long double)e / (long double)N);
    }
  }
  unsigned long long s = 12345;
  auto rnd = [&]() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(long long)(s >> 11) / 9007199254740992.0 - 0.5;
  };
  // sparse polynomial: the exact maximum over all N evaluation points
  {
    std::vector<double> f(N, 0.0);
    for (int t = 0; t < 40; t++)
      f[(size_t)((s = s * 6364136223846793005ull + 1442695040888963407ull) >> 40) % N] = rnd() * 1000.0;
    long double want = 0;
    for (unsigned j = 0; j < N; j++) {
      const long double v = direct_at(f, j);
      want = v > want ? v : want;
    }
    const double got = replay(f, wtab);
    if (!(std::fabs((long double)got - want) <= 1e-9L * want)) {
      printf("norm_replay FAILED (sparse): got %.17g want %.17Lg\n", got, want);
      return 1;
    }
  }
  // dense polynomial: the kernel's maximum must dominate every sampled point, and be attained (to 1e-9) at the
  // point a plain O(N^2) scan over a coarse subset cannot miss... so scan ALL points of a second, shorter check:
  {
    std::vector<double> f(N);
    for (auto& v : f)
      v = rnd();
    const double got = replay(f, wtab);
    long double best = 0;
    for (unsigned j = 0; j < N; j += 1) {   // full scan: 2.7e8 terms, a few seconds
      const long double v = direct_at(f, j);
      best = v > best ? v : best;
    }
    if (!(std::fabs((long double)got - best) <= 1e-9L * best)) {
      printf("norm_replay FAILED (dense): got %.17g want %.17Lg\n", got, best);
      return 1;
    }
  }
  // ---- N = 2^15: two sub-transforms ----
  {
    const unsigned N2 = 1u << 15;
    std::vector<tw16> w2(N2);
    std::vector<double> c2(2 * N2), s2(2 * N2);
    for (unsigned e = 0; e < 2 * N2; e++) {
      const long double ang = two_pi * (long double)e / (long double)(2 * N2);
      c2[e] = (double)cosl(ang);
      s2[e] = (double)sinl(ang);
      if (e < N2)
        w2[e] = {c2[e], s2[e]};
    }
    auto scan = [&](const std::vector<double>& f) {
      double best = 0;
      std::vector<unsigned> nz;
      for (unsigned i = 0; i < N2; i++)
        if (f[i] != 0.0)
          nz.push_back(i);
      for (unsigned j = 0; j < N2; j++) {
        double sr = 0, si = 0;
        for (unsigned i : nz) {
          const unsigned long e = ((unsigned long)i * (2ul * j + 1ul)) & (2ul * N2 - 1);
          sr += f[i] * c2[e];
          si += f[i] * s2[e];
        }
        const double v = sr * sr + si * si;
        best = v > best ? v : best;
      }
      return std::sqrt(best);
    };
    std::vector<double> f(N2, 0.0);
    for (int t = 0; t < 60; t++)
      f[(size_t)((s = s * 6364136223846793005ull + 1442695040888963407ull) >> 40) % N2] = rnd() * 1000.0;
    double want = scan(f), got = replay_x2(f, w2);
    if (!(std::fabs(got - want) <= 1e-9 * want)) {
      printf("norm_replay FAILED (N = 2^15 x2, sparse): got %.17g want %.17g\n", got, want);
      return 1;
    }
    for (auto& v : f)
      v = rnd();
    want = scan(f);
    got = replay_x2(f, w2);
    if (!(std::fabs(got - want) <= 1e-9 * want)) {
      printf("norm_replay FAILED (N = 2^15 x2, dense): got %.17g want %.17g\n", got, want);
      return 1;
    }
  }
  printf("norm_replay OK\n");
  return 0;
}

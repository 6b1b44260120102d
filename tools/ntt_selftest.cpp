// ntt_selftest.cpp -- standalone check of the NTT kernels through the C ABI against the CPU
// replay of the same phase functions (tests/cpp/ntt_replay.cpp).  No python, no torch.
//   g++ -O2 -std=c++17 tools/ntt_selftest.cpp tests/cpp/ntt_replay.cpp -Iinclude -ldl -o tools/ntt_selftest
//   ./tools/ntt_selftest path/to/libhelib_amd.so
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "helib_amd.h"

extern "C" int ntt_replay(int logn, int inverse, uint64_t q, uint64_t psi, const uint64_t* in,
                          uint64_t* out);

#define LOAD(name) auto p_##name = (decltype(&name))dlsym(h, #name); if (!p_##name) { printf("missing %s\n", #name); return 2; }

int main(int argc, char** argv)
{
  const char* so = argc > 1 ? argv[1] : "helib_amd/lib/libhelib_amd.so";
  void* h = dlopen(so, RTLD_NOW);
  if (!h) { printf("dlopen: %s\n", dlerror()); return 2; }
  LOAD(hx_ctx_create) LOAD(hx_ctx_add_prime) LOAD(hx_ctx_prime) LOAD(hx_poly_create) LOAD(hx_poly_upload)
  LOAD(hx_poly_download) LOAD(hx_ntt_forward) LOAD(hx_ntt_inverse) LOAD(hx_last_error) LOAD(hx_ctx_sync)
  // primes: q = 2^k t m + 1 found by trial (library validates primality and picks the root)
  int bad = 0;
  for (int logn = 13; logn <= 15; logn++) {
    uint64_t N = 1ull << logn, m = 2 * N;
    hx_ctx* ctx;
    if (p_hx_ctx_create(&ctx, 0, m)) { printf("ctx: %s\n", p_hx_last_error()); return 1; }
    int idx = -1;
    uint64_t q = 0, root = 0;
    for (uint64_t t = (1ull << 59) / m | 1;; t += 2) {
      uint64_t cand = t * m + 1;
      if (cand >> 60) { printf("no prime\n"); return 1; }
      if (p_hx_ctx_add_prime(ctx, cand, 0, &idx) == 0) { q = cand; break; }
    }
    p_hx_ctx_prime(ctx, idx, &q, &root);
    const int B = 3;
    hx_poly* poly;
    if (p_hx_poly_create(ctx, B, &idx, 1, &poly)) { printf("poly: %s\n", p_hx_last_error()); return 1; }
    std::vector<uint64_t> x(B * N), y(B * N), want(N), back(B * N);
    uint64_t s = 88172645463325252ull;
    for (auto& v : x) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s % q; }
    p_hx_poly_upload(poly, x.data());
    if (p_hx_ntt_forward(poly)) { printf("fwd: %s\n", p_hx_last_error()); return 1; }
    if (p_hx_poly_download(poly, y.data())) { printf("dl: %s\n", p_hx_last_error()); return 1; }
    for (int b = 0; b < B; b++) {
      ntt_replay(logn, 0, q, root, x.data() + b * N, want.data());
      size_t diff = 0;
      for (uint64_t j = 0; j < N; j++) diff += want[j] != y[b * N + j];
      printf("logn=%d b=%d forward mismatches=%zu\n", logn, b, diff);
      bad += diff != 0;
    }
    if (p_hx_ntt_inverse(poly)) { printf("inv: %s\n", p_hx_last_error()); return 1; }
    p_hx_poly_download(poly, back.data());
    size_t diff = 0;
    for (size_t j = 0; j < x.size(); j++) diff += back[j] != x[j];
    printf("logn=%d inverse round-trip mismatches=%zu\n", logn, diff);
    bad += diff != 0;
    fflush(stdout);
  }
  printf(bad ? "SELFTEST FAILED\n" : "SELFTEST OK\n");
  return bad ? 1 : 0;
}

// bench_cpp.cpp -- the BASELINE metric driven from the C++ host side (include/helib_amd_ctxt.hpp):
// Ctxt::multiplyBy on fresh ciphertexts at BGV m=32768, p=65537, bits=950, a batch of independent
// ciphertext pairs per step, operand copies made before the timer starts (benchmarks/bgv_basic.cpp:158-164;
// the engine's copies are copy-on-write, so nothing moves either way).  Synthetic uniform rows.  One JSON line.
//   g++ -O2 -std=c++17 -Iinclude tools/bench_cpp.cpp -Lhelib_amd/lib -lhelib_amd -Wl,-rpath,$PWD/helib_amd/lib -o bench_cpp
//   ./bench_cpp [batch=128] [steps=20] [warmup=3] [measure=0|1|2]     2: measured noise, norms read back lazily
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "helib_amd_ctxt.hpp"

using namespace helib_amd;

static uint64_t sm64(uint64_t& s)
{
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static std::vector<uint64_t> uniform(const ChainContext& cc, const IndexSet& idx, size_t batch, uint64_t& seed)
{
  size_t N = (size_t)cc.phim;
  std::vector<uint64_t> v(idx.size() * batch * N);
  size_t o = 0;
  for (int i : idx) {
    uint64_t q = cc.primes[(size_t)i];
    for (size_t j = 0; j < batch * N; j++)
      v[o++] = (uint64_t)(((unsigned __int128)sm64(seed) * q) >> 64);
  }
  return v;
}

int main(int argc, char** argv)
{
  int B = argc > 1 ? atoi(argv[1]) : 128, steps = argc > 2 ? atoi(argv[2]) : 20, warm = argc > 3 ? atoi(argv[3]) : 3;
  const int mode = argc > 4 ? atoi(argv[4]) : 0;
  const bool measure = mode != 0;
  Ctxt::deferNorms() = mode == 2;   // (LazyLn: a dropped result's estimate is never waited for, as in the python mirror)
  try {
    ChainContext cc(32768, 65537, 1, 950, 3);
    auto dev = cc.makeDeviceContext(0);
    IndexSet allp = cc.ctxtPrimes;
    allp.insert(allp.end(), cc.specialPrimes.begin(), cc.specialPrimes.end());
    uint64_t seed = 1234;
    size_t D = cc.digits.size();
    std::vector<uint64_t> kb, ka;
    for (size_t d = 0; d < D; d++) {
      auto b = uniform(cc, allp, 1, seed), a = uniform(cc, allp, 1, seed);
      kb.insert(kb.end(), b.begin(), b.end());
      ka.insert(ka.end(), a.begin(), a.end());
    }
    KeySwitch W(*dev, (int)D, allp, kb, ka);
    KeySet keys;
    keys.relin = &W;
    keys.ptxtSpace = cc.ptxtSpace;
    keys.lnNoise = std::log(cc.gaussBound() * cc.ptxtSpace);
    auto mk = [&]() {
      DoubleCRT d(*dev, cc.ctxtPrimes, B);
      d.setRows(uniform(cc, cc.ctxtPrimes, (size_t)B, seed));
      return d;
    };
    Ctxt fa = Ctxt::fresh(cc, *dev, keys, mk(), mk()), fb = Ctxt::fresh(cc, *dev, keys, mk(), mk());
    fa.measure = fb.measure = measure;
    auto run = [&](int k) {
      std::vector<Ctxt> as(k, fa), bs(k, fb);  // copies, untimed
      dev->sync();
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < k; i++)
        as[i].multiplyBy(std::move(bs[i]));  // both operand copies were made outside the timed region
      dev->sync();
      return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    run(std::max(warm, std::min(8, steps)));  // a full-size round first: the slab pool then holds every size the timed rounds ask for
    double dt = 0;
    int done = 0;
    while (done < steps) {
      int k = std::min(8, steps - done);
      dt += run(k);
      done += k;
    }
    printf("{\"metric\": \"ctxt_x_ctxt_mults_per_sec_incl_relinearize\", \"host\": \"C++ (helib_amd_ctxt.hpp)\", "
           "\"value\": %.1f, \"unit\": \"mult/s\", \"batch\": %d, \"steps\": %d, \"ms_per_step\": %.4f, "
           "\"noise\": \"%s\", \"workload\": \"BGV m=32768 p=65537 bits=950 L=%zu K=%zu D=%zu fresh multiplyBy\"}\n",
           (double)B * steps / dt, B, steps, dt / steps * 1e3, mode == 2 ? "measured (lazy read-back)" : measure ? "measured (synchronous read-back)" : "bounds",
           cc.ctxtPrimes.size(), cc.specialPrimes.size(), D);
  } catch (const std::exception& e) {
    fprintf(stderr, "exception: %s\n", e.what());
    return 1;
  }
  return 0;
}

// CPU-only check of the C++ host logic (include/helib_amd_ctxt.hpp): prints the prime chain, the
// digits, the ModuliSizes table size and the prime-set decision for two fresh ciphertexts as one
// JSON line; tests/test_host_logic.py compares it with the python mirror (helib_amd/ctxt.py).
//   g++ -std=c++17 -Iinclude tests/cpp/chain_test.cpp -Lhelib_amd/lib -lhelib_amd -o chain_test
#include <cstdio>
#include <cstdlib>

#include "helib_amd_ctxt.hpp"

using namespace helib_amd;

static void list(const char* name, const IndexSet& v, bool last = false)
{
  printf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); i++)
    printf("%s%d", i ? ", " : "", v[i]);
  printf("]%s", last ? "" : ", ");
}

int main(int argc, char** argv)
{
  if (argc < 4)
    return 2;
  long m = atol(argv[1]), p = atol(argv[2]), bits = atol(argv[3]);
  const bool ckks = p == -1;   // ContextBuilder<CKKS>().precision(20)
  ChainContext c(m, p, ckks ? 20 : 1, bits, 3, 3.2, 10.0, 0, 3, 0, ckks);
  printf("{\"primes\": [");
  for (size_t i = 0; i < c.primes.size(); i++)
    printf("%s%llu", i ? ", " : "", (unsigned long long)c.primes[i]);
  printf("], ");
  list("small", c.smallPrimes);
  list("ctxt", c.ctxtPrimes);
  list("special", c.specialPrimes);
  printf("\"digits\": [");
  for (size_t d = 0; d < c.digits.size(); d++) {
    printf("%s[", d ? ", " : "");
    for (size_t i = 0; i < c.digits[d].size(); i++)
      printf("%s%d", i ? ", " : "", c.digits[d][i]);
    printf("]");
  }
  printf("], \"nsizes\": %zu, \"fresh_ln\": %.17g, ", c.modSizes.count(), std::log(c.freshNoiseBound()));
  printf("\"bitSizeOfQ\": %ld, \"securityLevel\": %.17g, ", c.bitSizeOfQ(), c.securityLevel());
  // two fresh ciphertexts (parts "1" and "s"): Ctxt::computeIntervalForMul by hand -- no device here
  PrimeSet fresh = toSet(c.ctxtPrimes);
  double lnNoise = std::log(c.freshNoiseBound());
  double msn = (1.0 + c.skBound()) * c.noiseBoundForUniform(c.ptxtSpace / 2.0, c.phim);
  double hi = c.logOfProduct(fresh) - std::max(lnNoise, 0.0) + std::log(msn) - Ctxt::safety;
  std::pair<double, double> iv{hi - 4 * std::log(2.0), hi};
  if (ckks) {   // the opposite end of the window (src/Ctxt.cpp:1637-1651)
    double lo = c.logOfProduct(fresh) - std::max(lnNoise, 0.0) + std::log(msn) + Ctxt::safety;
    iv = {lo, lo + 4 * std::log(2.0)};
  }
  PrimeSet s = c.modSizes.getSet4Size(iv.first, iv.second, fresh, &fresh, ckks);
  printf("\"lo\": %.17g, \"hi\": %.17g, ", iv.first, iv.second);
  list("common", toVec(s), true);
  printf("}\n");
  return 0;
}

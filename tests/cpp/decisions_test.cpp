// CPU-only: ModuliSizes::init / getSet4Size of include/helib_amd_ctxt.hpp on
// the hand-derived cases of tests/golden/host_decisions.json (fed by tests/test_host_decisions_pinned.py
// as text lines on stdin; one answer line per case).  No device call is made.
//   table <pow2> <nprimes> q... <nsmall> i... <nctxt> i...
//   set4 <low> <high> <reverse> <n1> i... <n2|-1> i...
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string>

#include "helib_amd_ctxt.hpp"

using namespace helib_amd;

int main()
{
  ChainContext c(128, 257, 1, 100);
  fhe_stats() = true;   // the decision statistics of src/primeChain.cpp:207-208, 288-289 are collected on the way
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line);
    std::string op;
    in >> op;
    auto ints = [&](auto& out) {
      long n;
      in >> n;
      for (long i = 0; i < n; i++) {
        int v;
        in >> v;
        out.insert(out.end(), v);
      }
      return n;
    };
    if (op == "table") {
      int pow2;
      long n;
      in >> pow2 >> n;
      c.pow2 = pow2 != 0;
      c.primes.clear();
      for (long i = 0; i < n; i++) {
        unsigned long long q;
        in >> q;
        c.primes.push_back(q);
      }
      c.smallPrimes.clear();
      c.ctxtPrimes.clear();
      ints(c.smallPrimes);
      ints(c.ctxtPrimes);
      c.modSizes.init(c);
      printf("%zu\n", c.modSizes.count());
    } else if (op == "set4") {
      double lo, hi;
      int rev;
      in >> lo >> hi >> rev;
      PrimeSet f1, f2;
      ints(f1);
      long n2 = ints(f2);
      PrimeSet s = c.modSizes.getSet4Size(lo, hi, f1, n2 >= 0 ? &f2 : nullptr, rev != 0);
      for (int i : s)
        printf("%d ", i);
      printf("\n");
    }
  }
  // the reference's instrumentation hooks (helib_amd_timing.hpp): statistics and a named timer
  {
    HELIB_AMD_NTIMER_START(decisions_test_tail);
    HELIB_AMD_NTIMER_STOP(decisions_test_tail);
  }
  const FHEtimer* t = getTimerByName("decisions_test_tail");
  fprintf(stderr, "timer %s calls %ld\n", t ? t->name : "?", t ? t->getNumCalls() : -1L);
  print_stats(std::cerr);
  return 0;
}

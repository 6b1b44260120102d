// CPU check of ZmStar / family1D in include/helib_amd_keys.hpp (no device call on these paths):
//   zmstar_test <m> <p> [candidates...]   ->  JSON {gens, ords (signed), ordP, full, bsgs, min, frob}
#include <cstdio>
#include <cstdlib>

#include "helib_amd_keys.hpp"

using namespace helib_amd;

static void jl(const char* name, const std::vector<long>& v, bool last = false)
{
  printf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); i++)
    printf("%s%ld", i ? ", " : "", v[i]);
  printf("]%s", last ? "" : ", ");
}
static void fam(const char* name, const ZmStar& z, KSStrategy kind, bool last = false)
{
  printf("\"%s\": [", name);
  for (long i = 0; i < z.numOfGens(); i++) {
    std::vector<long> f = family1D(z, i, kind);
    printf("%s[", i ? ", " : "");
    for (size_t k = 0; k < f.size(); k++)
      printf("%s%ld", k ? ", " : "", f[k]);
    printf("]");
  }
  printf("]%s", last ? "" : ", ");
}

int main(int argc, char** argv)
{
  if (argc < 3)
    return 2;
  long m = atol(argv[1]), p = atol(argv[2]);
  std::vector<long> cand;
  for (int i = 3; i < argc; i++)
    cand.push_back(atol(argv[i]));
  ZmStar z(m, p, cand);
  printf("{");
  jl("gens", z.gens);
  jl("ords", z.signedOrds());
  printf("\"ordP\": %ld, \"nslots\": %ld, ", z.ordP, z.getNSlots());
  fam("full", z, HELIB_KSS_FULL);
  fam("bsgs", z, HELIB_KSS_BSGS);
  fam("min", z, HELIB_KSS_MIN);
  jl("frob", family1D(z, -1, HELIB_KSS_FULL), true);
  printf("}\n");
  return 0;
}

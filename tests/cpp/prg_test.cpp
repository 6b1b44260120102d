// CPU-only: the ChaCha20 generator of include/helib_amd_keys.hpp -- the RFC 8439 section 2.3.2 block
// (key 00..1f, counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00), the first outputs of a seeded
// generator (determinism) and that two entropy-keyed generators differ.  No device call.
#include <cstdio>

#include "helib_amd_keys.hpp"

using namespace helib_amd;

int main()
{
  uint8_t key[32];
  for (int i = 0; i < 32; i++)
    key[i] = (uint8_t)i;
  uint32_t out[16];
  ChaChaRng::block(key, 1, 0x09000000u, 0x4a000000u, 0, out);
  printf("block");
  for (int i = 0; i < 16; i++)
    printf(" %08x", out[i]);
  printf("\n");
  ChaChaRng a(42), b(42), c(43), e1, e2;
  const bool same = a() == b() && a() == b();
  const bool other = a() != c();
  printf("seeded %d %d\n", same, other);
  bool differ = false;
  for (int i = 0; i < 4; i++)
    differ = differ || e1() != e2();
  printf("entropy %d\n", differ);
  // a uniform draw through the standard distributions works (UniformRandomBitGenerator)
  std::uniform_int_distribution<uint64_t> u(0, 999);
  uint64_t v = u(a);
  printf("urbg %d\n", v < 1000);
  return 0;
}

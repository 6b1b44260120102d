// arena_test.cpp -- helib_amd/csrc/arena.h on the CPU (malloc stands in for hipMalloc): random
// allocate / release sequences never hand out overlapping extents, released extents coalesce back to
// whole chunks, steady-state loops stop reaching the system allocator, and the graph rules hold
// (blocks live during a capture are parked on release, blocks of eager work recycle).
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <vector>

#include "../../helib_amd/csrc/arena.h"

static size_t g_sys_allocs = 0, g_sys_frees = 0, g_sys_bytes = 0;
static size_t g_fail_above = ~(size_t)0;
static int sys_alloc(size_t bytes, void** out)
{
  if (g_sys_bytes + bytes > g_fail_above)
    return 2;  // hipErrorOutOfMemory
  // reserve address space only: the test never touches the memory
  *out = malloc(64);
  static uintptr_t next = (uintptr_t)1 << 40;
  // fake, non-overlapping "device" addresses so that huge chunks cost nothing on the host
  free(*out);
  *out = (void*)next;
  next += (bytes + ((size_t)1 << 30)) & ~(((size_t)1 << 30) - 1);
  g_sys_allocs++;
  g_sys_bytes += bytes;
  return 0;
}
static void sys_free(void*) { g_sys_frees++; }

#define REQUIRE(x)                                                       \
  do {                                                                   \
    if (!(x)) {                                                          \
      printf("arena_test FAILED at line %d: %s\n", __LINE__, #x);        \
      return 1;                                                          \
    }                                                                    \
  } while (0)

int main()
{
  using hxa::SlabArena;
  const size_t G = SlabArena::GRAIN;
  {
    SlabArena a;
    a.sys_alloc = sys_alloc;
    a.sys_free = sys_free;
    std::mt19937_64 rng(12345);
    std::vector<std::pair<char*, size_t>> held;
    auto overlaps = [&](char* p, size_t len) {
      for (auto& h : held)
        if (p < h.first + h.second && h.first < p + len)
          return true;
      return false;
    };
    for (int it = 0; it < 20000; it++) {
      const bool do_alloc = held.empty() || (rng() % 100) < (held.size() < 100 ? 60u : 40u);
      if (do_alloc) {
        size_t bytes = (size_t)(1 + rng() % 300) * G - (rng() % G);
        void* p = nullptr;
        REQUIRE(a.alloc(bytes, false, &p) == 0);
        REQUIRE(((uintptr_t)p % G) == 0);
        REQUIRE(!overlaps((char*)p, SlabArena::round_up(bytes)));
        held.emplace_back((char*)p, SlabArena::round_up(bytes));
      } else {
        size_t i = rng() % held.size();
        a.release(held[i].first);
        held[i] = held.back();
        held.pop_back();
      }
      size_t sum = 0;
      for (auto& h : held)
        sum += h.second;
      REQUIRE(sum == a.in_use);
    }
    for (auto& h : held)
      a.release(h.first);
    REQUIRE(a.in_use == 0);
    // everything coalesced: one free extent per chunk, of the chunk's size
    size_t nch = 0;
    for (auto& ch : a.chunks) {
      if (!ch.base)
        continue;
      nch++;
      REQUIRE(ch.free.size() == 1 && ch.free.begin()->first == 0 && ch.free.begin()->second == ch.size);
    }
    REQUIRE(a.by_size.size() == nch);
    // geometric growth: tens of GB of peak demand in a handful of system calls
    REQUIRE(a.sys_calls <= 16);
    const size_t before = a.reserved;
    REQUIRE(a.trim(0) == before && a.reserved == 0);
    a.destroy();
  }
  {
    // a benchmark loop that keeps its results alive: after warm-up no system allocation per step
    SlabArena a;
    a.sys_alloc = sys_alloc;
    a.sys_free = sys_free;
    const size_t slab = 18 * 128 * 16384 * 8;  // one DoubleCRT of the bench: 18 rows x batch 128 x N
    std::vector<void*> results;
    for (int step = 0; step < 3; step++) {       // warm-up rounds
      for (int i = 0; i < 9; i++) {
        void *t1, *t2, *r;
        REQUIRE(a.alloc(slab, false, &t1) == 0 && a.alloc(slab, false, &t2) == 0 && a.alloc(slab, false, &r) == 0);
        a.release(t1);
        a.release(t2);
        results.push_back(r);
      }
      for (void* r : results)
        a.release(r);
      results.clear();
    }
    const size_t calls = a.sys_calls;
    for (int step = 0; step < 50; step++) {
      for (int i = 0; i < 9; i++) {
        void *t1, *t2, *r;
        REQUIRE(a.alloc(slab, false, &t1) == 0 && a.alloc(slab, false, &t2) == 0 && a.alloc(slab, false, &r) == 0);
        a.release(t1);
        a.release(t2);
        results.push_back(r);
      }
      for (void* r : results)
        a.release(r);
      results.clear();
    }
    REQUIRE(a.sys_calls == calls);
    a.destroy();
  }
  {
    // HIP-graph rules
    SlabArena a;
    a.sys_alloc = sys_alloc;
    a.sys_free = sys_free;
    void *in, *tmp, *eager;
    REQUIRE(a.alloc(10 * G, false, &in) == 0);
    a.pin_all();                                   // capture begins: `in` may be referenced
    REQUIRE(a.alloc(10 * G, true, &tmp) == 0);     // taken during the capture
    REQUIRE(a.is_pinned(in) && a.is_pinned(tmp));
    a.defer(tmp);                                  // its poly dies while the graph lives: parked
    const size_t use0 = a.in_use;
    // eager work next to the live graph: recycles, no growth
    const size_t res0 = a.reserved;
    for (int i = 0; i < 1000; i++) {
      REQUIRE(a.alloc(10 * G, false, &eager) == 0);
      REQUIRE(!a.is_pinned(eager) && eager != tmp && eager != in);
      a.release(eager);
    }
    REQUIRE(a.reserved == res0 || a.reserved <= res0 + 64 * G);
    REQUIRE(a.in_use == use0);
    a.unpin_all();                                 // last graph destroyed
    REQUIRE(!a.is_pinned(in) && a.deferred.empty() && a.in_use == 10 * G);
    a.release(in);
    REQUIRE(a.in_use == 0);
    a.destroy();
  }
  {
    // out of memory: the error of the system allocator comes back; after releasing, allocation works
    SlabArena a;
    a.sys_alloc = sys_alloc;
    a.sys_free = sys_free;
    g_fail_above = g_sys_bytes + 200 * G;
    void *p1 = nullptr, *p2 = nullptr;
    REQUIRE(a.alloc(150 * G, false, &p1) == 0);
    REQUIRE(a.alloc(150 * G, false, &p2) != 0);
    a.release(p1);
    REQUIRE(a.alloc(100 * G, false, &p2) == 0);
    g_fail_above = ~(size_t)0;
    a.destroy();
  }
  printf("arena_test OK\n");
  return 0;
}

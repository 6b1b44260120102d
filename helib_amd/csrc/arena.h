// arena.h -- device-memory arena behind every DoubleCRT slab of a context (host code only).
//
// HElib allocates one NTL vec_long per row (include/helib/DoubleCRT.h:87-95) and lets malloc
// recycle them.  On the GPU a hipMalloc of a 0.5 GB slab costs milliseconds (and has a slow mode of
// tens of milliseconds on part of the MI355X pool), so a benchmark loop that keeps its results
// alive -- benchmarks/bgv_basic.cpp:36-211 does -- must not reach hipMalloc per operation.  The
// arena takes a few large chunks from the system allocator (geometric growth, sized for a 288 GB
// part) and sub-allocates slabs from them: best fit over the free extents, neighbours coalesced on
// release.  Every slab of one context is used on that context's single stream, so an extent
// released by the host may be handed out again at once -- reuse is ordered by the stream.
//
// HIP graphs: a captured graph has device addresses baked in.  Blocks that were live at any time
// while a capture was open are `pinned`; releasing a pinned block while a graph may still replay
// parks it in `deferred` instead of the free lists (engine.hip decides when: capturing or graphs
// alive).  Blocks allocated by eager work outside a capture are never pinned and recycle normally,
// so eager work next to a live graph does not grow device memory.
//
// The system allocator is a pair of function pointers so that the logic is unit-tested on the CPU
// (tests/cpp/arena_test.cpp) with malloc/free.
#pragma once
#include <cstddef>
#include <cstdint>
#include <map>
#include <unordered_map>
#include <utility>
#include <vector>

namespace hxa {

struct SlabArena {
  using AllocFn = int (*)(size_t bytes, void** out);  // 0 on success
  using FreeFn = void (*)(void* p);
  static constexpr size_t GRAIN = (size_t)2 << 20;  // extents are multiples of 2 MiB

  AllocFn sys_alloc = nullptr;
  FreeFn sys_free = nullptr;
  size_t first_chunk = (size_t)64 << 20;  // a small ring's context stays small
  size_t max_chunk = (size_t)8 << 30;

  struct Chunk {
    char* base = nullptr;  // nullptr: slot of a chunk that was given back
    size_t size = 0, used = 0;
    std::map<size_t, size_t> free;  // offset -> length
  };
  struct Block {
    int chunk;
    size_t off, len;
    bool pinned;
  };
  std::vector<Chunk> chunks;
  std::multimap<size_t, std::pair<int, size_t>> by_size;  // length -> (chunk, offset)
  std::unordered_map<void*, Block> live;
  std::vector<void*> deferred;
  size_t reserved = 0, in_use = 0, sys_calls = 0;

  static size_t round_up(size_t bytes) { return bytes == 0 ? GRAIN : (bytes + GRAIN - 1) / GRAIN * GRAIN; }

  // 0 on success; otherwise the system allocator's error for a chunk of at least `bytes`
  int alloc(size_t bytes, bool pinned, void** out)
  {
    const size_t len = round_up(bytes);
    auto it = by_size.lower_bound(len);
    if (it == by_size.end()) {
      int rc = grow(len);
      if (rc != 0)
        return rc;
      it = by_size.lower_bound(len);
    }
    const size_t flen = it->first;
    const int ci = it->second.first;
    const size_t off = it->second.second;
    by_size.erase(it);
    Chunk& ch = chunks[(size_t)ci];
    ch.free.erase(off);
    if (flen > len)
      put_free(ci, off + len, flen - len);
    ch.used += len;
    in_use += len;
    void* p = ch.base + off;
    live[p] = Block{ci, off, len, pinned};
    *out = p;
    return 0;
  }
  bool is_pinned(void* p) const
  {
    auto it = live.find(p);
    return it != live.end() && it->second.pinned;
  }
  bool owns(void* p) const { return live.find(p) != live.end(); }
  // back to the free lists (coalescing with both neighbours)
  void release(void* p)
  {
    auto it = live.find(p);
    if (it == live.end())
      return;
    const Block b = it->second;
    live.erase(it);
    chunks[(size_t)b.chunk].used -= b.len;
    in_use -= b.len;
    size_t off = b.off, len = b.len;
    Chunk& ch = chunks[(size_t)b.chunk];
    auto next = ch.free.lower_bound(off);
    if (next != ch.free.end() && off + len == next->first) {
      len += next->second;
      drop_size(b.chunk, next->first, next->second);
      next = ch.free.erase(next);
    }
    if (next != ch.free.begin()) {
      auto prev = std::prev(next);
      if (prev->first + prev->second == off) {
        off = prev->first;
        len += prev->second;
        drop_size(b.chunk, prev->first, prev->second);
        ch.free.erase(prev);
      }
    }
    put_free(b.chunk, off, len);
  }
  // a pinned block whose owner is gone but which a graph may still read or write
  void defer(void* p) { deferred.push_back(p); }
  void pin_all()
  {
    for (auto& kv : live)
      kv.second.pinned = true;
  }
  // no graph left: nothing is pinned any more and the parked blocks are free again
  void unpin_all()
  {
    for (auto& kv : live)
      kv.second.pinned = false;
    std::vector<void*> d;
    d.swap(deferred);
    for (void* p : d)
      release(p);
  }
  // make sure at least `bytes` are reserved from the system allocator (one chunk for the difference), so that
  // a loop whose footprint is known up front never reaches the system allocator while it is being timed
  int reserve(size_t bytes)
  {
    if (reserved >= bytes)
      return 0;
    return grow_exact(round_up(bytes - reserved));
  }
  size_t cached() const { return reserved - in_use; }
  // give wholly free chunks back to the system until at most `keep` bytes stay cached (the caller has
  // drained the stream: the system allocator is not stream-ordered); returns the bytes released
  size_t trim(size_t keep)
  {
    size_t freed = 0;
    for (size_t i = chunks.size(); i-- > 0 && cached() > keep;) {
      Chunk& ch = chunks[i];
      if (!ch.base || ch.used != 0)
        continue;
      for (auto& kv : ch.free)
        drop_size((int)i, kv.first, kv.second);
      ch.free.clear();
      sys_free(ch.base);
      reserved -= ch.size;
      freed += ch.size;
      ch.base = nullptr;
      ch.size = 0;
    }
    return freed;
  }
  // everything goes (context teardown)
  void destroy()
  {
    for (Chunk& ch : chunks)
      if (ch.base)
        sys_free(ch.base);
    chunks.clear();
    by_size.clear();
    live.clear();
    deferred.clear();
    reserved = in_use = 0;
  }

 private:
  void put_free(int ci, size_t off, size_t len)
  {
    chunks[(size_t)ci].free[off] = len;
    by_size.emplace(len, std::make_pair(ci, off));
  }
  void drop_size(int ci, size_t off, size_t len)
  {
    auto r = by_size.equal_range(len);
    for (auto it = r.first; it != r.second; ++it)
      if (it->second.first == ci && it->second.second == off) {
        by_size.erase(it);
        return;
      }
  }
  int grow_exact(size_t want)
  {
    void* p = nullptr;
    sys_calls++;
    int rc = sys_alloc(want, &p);
    if (rc != 0)
      return rc;
    add_chunk(p, want);
    return 0;
  }
  void add_chunk(void* p, size_t want)
  {
    int ci = -1;
    for (size_t i = 0; i < chunks.size(); i++)
      if (!chunks[i].base) {
        ci = (int)i;
        break;
      }
    if (ci < 0) {
      chunks.emplace_back();
      ci = (int)chunks.size() - 1;
    }
    chunks[(size_t)ci].base = (char*)p;
    chunks[(size_t)ci].size = want;
    chunks[(size_t)ci].used = 0;
    reserved += want;
    put_free(ci, 0, want);
  }
  int grow(size_t len)
  {
    // geometric: the next chunk is as large as everything reserved so far (at least first_chunk, at
    // most max_chunk) -- a dozen system allocations take a context to tens of GB
    size_t want = reserved < first_chunk ? first_chunk : reserved;
    if (want > max_chunk)
      want = max_chunk;
    if (want < len)
      want = len;
    void* p = nullptr;
    sys_calls++;
    int rc = sys_alloc(want, &p);
    if (rc != 0 && want > len) {
      want = len;
      sys_calls++;
      rc = sys_alloc(want, &p);
    }
    // (no trim here: giving empty chunks back means the system free, a device-wide wait that must not happen inside
    // a stream capture -- the owner decides: engine.hip's pool_alloc drains its stream, trims and retries)
    if (rc != 0)
      return rc;
    add_chunk(p, want);
    return 0;
  }
};

}  // namespace hxa

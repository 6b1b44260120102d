// prof.h -- in-situ kernel timing with HIP events (host code only).
//
// Every kernel launch of the library goes through HX_LAUNCH.  Between hx_profile_begin() and
// hx_profile_end() each launch carries a pair of events that take the dispatch's own start and stop
// timestamps (hipExtLaunchKernelGGL: no extra barrier or cache write-back between kernels, which a
// hipEventRecord pair would insert -- measured: it made the mod-down apply kernel look 8 % slower than
// rocprofv3 sees it), so a kernel is timed where it actually runs -- between its real neighbours inside
// a multiply, with their cache and clock state -- and not in a back-to-back loop of its own.
// bench.py derives `roofline` from this (the rocprofv3 kernel trace of the same command, committed
// under profiles/, is the cross-check); outside a profiling window a launch costs one predictable
// branch.  Launches recorded into a HIP graph are skipped (events cannot bracket them).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cxxabi.h>
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace hxp {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: raise it once per (device, kernel),
// under a lock -- a process-wide `static bool` (rounds 1-5) left a second context on another device with the default
// 64 KiB limit (its first large-LDS launch failed) and let two contexts' threads race on the flag.
inline hipError_t dyn_lds(const void* kern, int bytes)
{
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> done;   // (device, kernel) -> bytes granted
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess)
    return e;
  std::lock_guard<std::mutex> lk(mu);
  auto it = done.find({dev, kern});
  if (it != done.end() && it->second >= bytes)
    return hipSuccess;
  e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess)
    done[{dev, kern}] = bytes;
  return e;
}
// (the signature of hipFuncSetAttribute, for the call sites that were written against it)
inline hipError_t dyn_lds(const void* kern, hipFuncAttribute, int bytes) { return dyn_lds(kern, bytes); }

struct Rec {
  int name, device;
  unsigned wgs, wg_size;
  hipEvent_t e0, e1;
};
struct State {
  std::mutex mu;
  std::vector<std::string> names;
  std::unordered_map<const void*, int> by_ptr;
  std::vector<Rec> recs;
  std::map<int, std::vector<hipEvent_t>> spare;   // by device: an event is reused on the device it was created on
  size_t dropped = 0;
  std::string summary;                            // of the window that ended last, until it has been fetched
  bool have_summary = false;
};
// read by every launch from any thread, written under State::mu by begin() / end()
inline std::atomic<bool> enabled{false};
inline State& state()
{
  static State s;
  return s;
}
static constexpr size_t MAX_RECS = (size_t)1 << 18;

inline hipEvent_t take_event(State& s, int device)
{
  hipEvent_t e = nullptr;
  std::vector<hipEvent_t>& pool = s.spare[device];
  if (!pool.empty()) {
    e = pool.back();
    pool.pop_back();
  } else if (hipEventCreate(&e) != hipSuccess) {
    (void)hipGetLastError();
    e = nullptr;
  }
  return e;
}

// registers the launch; true: launch with (e0, e1) as the dispatch's start / stop events
inline bool pre(const void* fn, const char* text, hipStream_t st, dim3 grid, dim3 block, hipEvent_t* e0, hipEvent_t* e1)
{
  State& s = state();
  std::lock_guard<std::mutex> lk(s.mu);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (cs != hipStreamCaptureStatusNone)
    return false;
  if (s.recs.size() >= MAX_RECS) {
    s.dropped++;
    return false;
  }
  auto it = s.by_ptr.find(fn);
  if (it == s.by_ptr.end()) {
    // the instantiated kernel's name as rocprofv3 prints it
    std::string nm;
    const char* mangled = hipKernelNameRefByPtr(fn, st);
    if (mangled) {
      int status = 0;
      char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &status);
      nm = (status == 0 && dem) ? dem : mangled;
      free(dem);
    } else {
      (void)hipGetLastError();
      nm = text;
    }
    const size_t paren = nm.find('(');  // drop the argument list
    if (paren != std::string::npos && paren > 0)
      nm.resize(paren);
    if (nm.rfind("void ", 0) == 0)
      nm.erase(0, 5);
    s.names.push_back(nm);
    it = s.by_ptr.emplace(fn, (int)s.names.size() - 1).first;
  }
  Rec r;
  r.name = it->second;
  r.device = 0;
  if (hipGetDevice(&r.device) != hipSuccess)
    (void)hipGetLastError();
  r.wgs = grid.x * grid.y * grid.z;
  r.wg_size = block.x * block.y * block.z;
  r.e0 = take_event(s, r.device);
  r.e1 = take_event(s, r.device);
  if (!r.e0 || !r.e1) {
    if (r.e0)
      s.spare[r.device].push_back(r.e0);
    if (r.e1)
      s.spare[r.device].push_back(r.e1);
    s.dropped++;
    return false;
  }
  s.recs.push_back(r);
  *e0 = r.e0;
  *e1 = r.e1;
  return true;
}

inline int begin()
{
  State& s = state();
  std::lock_guard<std::mutex> lk(s.mu);
  for (Rec& r : s.recs) {
    s.spare[r.device].push_back(r.e0);
    s.spare[r.device].push_back(r.e1);
  }
  s.recs.clear();
  s.dropped = 0;
  s.summary.clear();
  s.have_summary = false;
  enabled.store(true, std::memory_order_relaxed);
  return 0;
}
// waits for the recorded launches; JSON: {"launches": n, "dropped": d, "kernels": [{"kernel", "workgroups",
// "workgroup_size", "calls", "total_us", "avg_us", "min_us", "max_us"} ... by total time]}
// The window closes at the FIRST call (recording stops there whatever becomes of the summary); the summary is kept
// -- under the mutex, for whichever thread asks -- until fetch = true hands it out or the next begin() drops it.
inline std::string end(bool fetch)
{
  State& s = state();
  std::lock_guard<std::mutex> lk(s.mu);
  enabled.store(false, std::memory_order_relaxed);
  if (s.have_summary) {
    std::string out = s.summary;
    if (fetch) {
      s.summary.clear();
      s.have_summary = false;
    }
    return out;
  }
  struct Agg {
    size_t calls = 0;
    double total = 0, mn = 1e30, mx = 0;
    unsigned wg_size = 0;
  };
  std::map<std::pair<int, unsigned>, Agg> agg;
  size_t bad = 0;
  for (Rec& r : s.recs) {
    float ms = 0;
    if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
      (void)hipGetLastError();
      bad++;
    } else {
      Agg& a = agg[{r.name, r.wgs}];
      const double us = (double)ms * 1e3;
      a.calls++;
      a.total += us;
      a.mn = us < a.mn ? us : a.mn;
      a.mx = us > a.mx ? us : a.mx;
      a.wg_size = r.wg_size;
    }
    s.spare[r.device].push_back(r.e0);
    s.spare[r.device].push_back(r.e1);
  }
  const size_t n = s.recs.size();
  s.recs.clear();
  std::vector<std::pair<double, std::pair<int, unsigned>>> order;
  for (auto& kv : agg)
    order.push_back({-kv.second.total, kv.first});
  std::sort(order.begin(), order.end());
  std::string out = "{\"launches\": " + std::to_string(n) + ", \"dropped\": " + std::to_string(s.dropped + bad) +
                    ", \"kernels\": [";
  bool first = true;
  for (auto& o : order) {
    const Agg& a = agg[o.second];
    std::string nm;
    for (char ch : s.names[(size_t)o.second.first])
      if (ch != '"' && ch != '\\')
        nm.push_back(ch);
    char buf[256];
    snprintf(buf, sizeof buf,
             "\"workgroups\": %u, \"workgroup_size\": %u, \"calls\": %zu, \"total_us\": %.2f, \"avg_us\": %.3f, "
             "\"min_us\": %.3f, \"max_us\": %.3f}",
             o.second.second, a.wg_size, a.calls, a.total, a.total / (double)a.calls, a.mn, a.mx);
    out += std::string(first ? "" : ", ") + "{\"kernel\": \"" + nm + "\", " + buf;
    first = false;
  }
  out += "]}";
  if (!fetch) {
    s.summary = out;
    s.have_summary = true;
  }
  return out;
}

}  // namespace hxp

#define HX_LAUNCH(kern, grid, block, lds, st, ...)                                                        \
  do {                                                                                                     \
    const dim3 _hx_g = (grid), _hx_b = (block);                                                            \
    hipEvent_t _hx_e0, _hx_e1;                                                                             \
    if (hxp::enabled.load(std::memory_order_relaxed) && hxp::pre((const void*)(kern), #kern, (st), _hx_g, _hx_b, &_hx_e0, &_hx_e1))        \
      hipExtLaunchKernelGGL(kern, _hx_g, _hx_b, (lds), (st), _hx_e0, _hx_e1, 0, __VA_ARGS__);              \
    else                                                                                                   \
      hipLaunchKernelGGL(kern, _hx_g, _hx_b, (lds), (st), __VA_ARGS__);                                    \
  } while (0)

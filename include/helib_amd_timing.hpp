// helib_amd_timing.hpp -- named timers and statistics of the C++ host side: the reference's
// instrumentation hooks for this path, same names and behaviour:
//   FHEtimer / auto_timer / HELIB_TIMER_START / HELIB_NTIMER_START / getTimerByName /
//   resetAllTimers / printAllTimers / printNamedTimer      include/helib/timing.h:44-131, src/timing.cpp
//   fhe_stats / fhe_stats_record / HELIB_STATS_UPDATE / HELIB_STATS_SAVE / print_stats /
//   fetch_saved_values                                      include/helib/fhe_stats.h:21-58, src/fhe_stats.cpp
// (macros carry the HELIB_AMD_ prefix so that both libraries can be seen by one translation unit).
// A timer accumulates HOST time: the engine enqueues its device work asynchronously, so a timer
// shows what the host spent issuing the call unless timers_sync_device() is set to a function that
// waits for the device (then every stop waits first and the timers show device-inclusive time).
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <iostream>
#include <mutex>
#include <vector>

namespace helib_amd {

class FHEtimer;
inline std::vector<FHEtimer*>& timerMap()
{
  static std::vector<FHEtimer*> m;
  return m;
}
inline std::mutex& timerMutex()
{
  static std::mutex mu;
  return mu;
}
inline void registerTimer(FHEtimer* t)
{
  std::lock_guard<std::mutex> g(timerMutex());
  timerMap().push_back(t);
}
inline unsigned long GetTimerClock()  // nanoseconds of a monotonic clock
{
  return (unsigned long)std::chrono::duration_cast<std::chrono::nanoseconds>(
             std::chrono::steady_clock::now().time_since_epoch())
      .count();
}
// optional: called before a timer stops (e.g. [&]{ ctx.sync(); })
inline void (*&timers_sync_device())()
{
  static void (*fn)() = nullptr;
  return fn;
}

class FHEtimer {
public:
  const char* name;
  const char* loc;
  std::atomic<unsigned long> counter;
  std::atomic<long> numCalls;
  FHEtimer(const char* n, const char* l) : name(n), loc(l), counter(0), numCalls(0) { registerTimer(this); }
  void reset()
  {
    counter = 0;
    numCalls = 0;
  }
  double getTime() const { return (double)counter.load() * 1e-9; }
  long getNumCalls() const { return numCalls.load(); }
};

inline void setTimersOn() {}
inline void setTimersOff() {}
inline bool areTimersOn() { return true; }

inline const FHEtimer* getTimerByName(const char* name)
{
  std::lock_guard<std::mutex> g(timerMutex());
  for (FHEtimer* t : timerMap())
    if (std::strcmp(t->name, name) == 0)
      return t;
  return nullptr;
}
inline void resetAllTimers()
{
  std::lock_guard<std::mutex> g(timerMutex());
  for (FHEtimer* t : timerMap())
    t->reset();
}
inline void printTimer(std::ostream& str, const FHEtimer& t)
{
  long n = t.getNumCalls();
  double ave = n > 0 ? t.getTime() / (double)n : 0.0;
  str << "  " << t.name << ": " << t.getTime() << " / " << n << " = " << ave << "   [" << t.loc << "]\n";
}
inline void printAllTimers(std::ostream& str = std::cerr)
{
  std::vector<FHEtimer*> v;
  {
    std::lock_guard<std::mutex> g(timerMutex());
    v = timerMap();
  }
  std::sort(v.begin(), v.end(), [](FHEtimer* a, FHEtimer* b) { return std::strcmp(a->name, b->name) < 0; });
  for (FHEtimer* t : v)
    printTimer(str, *t);
}
inline bool printNamedTimer(std::ostream& str, const char* name)
{
  const FHEtimer* t = getTimerByName(name);
  if (!t)
    return false;
  printTimer(str, *t);
  return true;
}

class auto_timer {
public:
  FHEtimer* timer;
  unsigned long amt;
  bool running;
  explicit auto_timer(FHEtimer* t) : timer(t), amt(GetTimerClock()), running(true) {}
  void stop()
  {
    if (!running)
      return;
    if (timers_sync_device())
      timers_sync_device()();
    amt = GetTimerClock() - amt;
    timer->counter += amt;
    timer->numCalls++;
    running = false;
  }
  ~auto_timer() { stop(); }
};

#define HELIB_AMD_STRINGIFY(x) #x
#define HELIB_AMD_TOSTRING(x) HELIB_AMD_STRINGIFY(x)
#define HELIB_AMD_AT __FILE__ ":" HELIB_AMD_TOSTRING(__LINE__)
#define HELIB_AMD_TIMER_START                                      \
  static helib_amd::FHEtimer _local_timer(__func__, HELIB_AMD_AT); \
  helib_amd::auto_timer _local_auto_timer(&_local_timer)
#define HELIB_AMD_TIMER_STOP _local_auto_timer.stop()
#define HELIB_AMD_NTIMER_START(n)                                        \
  static helib_amd::FHEtimer _named_local_timer##n(#n, HELIB_AMD_AT);    \
  helib_amd::auto_timer _named_local_auto_timer##n(&_named_local_timer##n)
#define HELIB_AMD_NTIMER_STOP(n) _named_local_auto_timer##n.stop()

// ---- statistics ----
inline bool& fhe_stats()
{
  static bool on = false;
  return on;
}
struct fhe_stats_record {
  const char* name;
  long count = 0;
  double sum = 0, max = 0;
  std::vector<double> saved_values;
  static std::vector<fhe_stats_record*>& map()
  {
    static std::vector<fhe_stats_record*> m;
    return m;
  }
  explicit fhe_stats_record(const char* n) : name(n)
  {
    std::lock_guard<std::mutex> g(timerMutex());
    map().push_back(this);
  }
  void update(double v)
  {
    std::lock_guard<std::mutex> g(timerMutex());
    count++;
    sum += v;
    if (v > max)
      max = v;
  }
  void save(double v)
  {
    std::lock_guard<std::mutex> g(timerMutex());
    saved_values.push_back(v);
  }
};
#define HELIB_AMD_STATS_UPDATE(name, val)                             \
  do {                                                                \
    if (helib_amd::fhe_stats()) {                                     \
      static helib_amd::fhe_stats_record _local_stats_record(name);   \
      _local_stats_record.update(val);                                \
    }                                                                 \
  } while (0)
#define HELIB_AMD_STATS_SAVE(name, val)                               \
  do {                                                                \
    if (helib_amd::fhe_stats()) {                                     \
      static helib_amd::fhe_stats_record _local_stats_record(name);   \
      _local_stats_record.save(val);                                  \
    }                                                                 \
  } while (0)
inline void print_stats(std::ostream& s)
{
  std::vector<fhe_stats_record*> v = fhe_stats_record::map();
  std::sort(v.begin(), v.end(),
            [](fhe_stats_record* a, fhe_stats_record* b) { return std::strcmp(a->name, b->name) < 0; });
  for (fhe_stats_record* r : v) {
    if (!r->saved_values.empty())
      s << r->name << " saved values: " << r->saved_values.size() << "\n";
    else
      s << r->name << " ave=" << (r->count ? r->sum / (double)r->count : 0.0) << " max=" << r->max << "\n";
  }
}
inline const std::vector<double>* fetch_saved_values(const char* name)
{
  for (fhe_stats_record* r : fhe_stats_record::map())
    if (std::strcmp(r->name, name) == 0)
      return &r->saved_values;
  return nullptr;
}

}  // namespace helib_amd

"""Named timers and statistics of the host side -- the reference's instrumentation hooks for this
path: FHEtimer / HELIB_TIMER_START / HELIB_NTIMER_START / getTimerByName / printAllTimers
(include/helib/timing.h:44-131, src/timing.cpp) and fhe_stats / HELIB_STATS_UPDATE / HELIB_STATS_SAVE
/ print_stats / fetch_saved_values (include/helib/fhe_stats.h:21-58, src/fhe_stats.cpp).

A timer accumulates HOST time between start and stop, as the reference's do.  The device work of
this engine is enqueued asynchronously, so by default a timer shows what the host spent issuing the
operation; set `sync_device = ctx.sync` (any callable) to make every stop wait for the device first
and the timers show elapsed device time per call site instead (slower: it serialises the pipeline).
"""
import functools
import sys
import time

_timers = {}          # name -> FHEtimer   (registerTimer)
sync_device = None    # optional callable run before a timer stops
fhe_stats = False     # helib::fhe_stats: statistics are collected only when set
_stats = {}           # name -> StatsRecord


class FHEtimer:
    """include/helib/timing.h:44-64: accumulated seconds and number of calls of one call site"""

    def __init__(self, name, loc=""):
        self.name, self.loc = name, loc
        self.counter = 0.0
        self.numCalls = 0

    def reset(self):
        self.counter, self.numCalls = 0.0, 0

    def getTime(self):
        return self.counter

    def getNumCalls(self):
        return self.numCalls


class auto_timer:
    """include/helib/timing.h:84-110 (also a context manager)"""

    def __init__(self, timer):
        self.timer, self.amt, self.running = timer, time.perf_counter(), True

    def stop(self):
        if self.running:
            if sync_device is not None:
                sync_device()
            self.timer.counter += time.perf_counter() - self.amt
            self.timer.numCalls += 1
            self.running = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.stop()
        return False


def _timer(name, loc=""):
    t = _timers.get(name)
    if t is None:
        t = _timers[name] = FHEtimer(name, loc)
    return t


def NTIMER_START(name, loc=""):
    """HELIB_NTIMER_START(name): `with NTIMER_START("KS_loop"): ...` or keep the object and .stop()"""
    return auto_timer(_timer(name, loc))


def timed(fn):
    """HELIB_TIMER_START at the top of a function: the timer is named after the function"""
    t = _timer(fn.__qualname__.split(".")[-1], f"{fn.__code__.co_filename}:{fn.__code__.co_firstlineno}")

    @functools.wraps(fn)
    def wrapper(*a, **k):
        at = auto_timer(t)
        try:
            return fn(*a, **k)
        finally:
            at.stop()
    return wrapper


def setTimersOn():      # backward compatibility in the reference too: timers are always on
    pass


def setTimersOff():
    pass


def areTimersOn():
    return True


def getTimerByName(name):
    return _timers.get(name)


def resetAllTimers():
    for t in _timers.values():
        t.reset()


def printNamedTimer(name, out=None):
    """src/timing.cpp:printNamedTimer: `name: total / calls = average   [loc]`; False if unknown"""
    t = _timers.get(name)
    if t is None:
        return False
    n = t.getNumCalls()
    ave = t.getTime() / n if n > 0 else 0.0
    (out or sys.stderr).write(f"  {t.name}: {t.getTime()} / {n} = {ave}   [{t.loc}]\n")
    return True


def printAllTimers(out=None):
    for name in sorted(_timers):
        printNamedTimer(name, out)


class StatsRecord:
    """include/helib/fhe_stats.h:21-36"""

    def __init__(self, name):
        self.name, self.count, self.sum, self.max = name, 0, 0.0, 0.0
        self.saved_values = []

    def update(self, val):
        val = float(val)
        self.count += 1
        self.sum += val
        if val > self.max:
            self.max = val

    def save(self, val):
        self.saved_values.append(float(val))


def STATS_UPDATE(name, val):
    """HELIB_STATS_UPDATE(name, val): count / sum / max of val, only while fhe_stats is set"""
    if fhe_stats:
        r = _stats.get(name)
        if r is None:
            r = _stats[name] = StatsRecord(name)
        r.update(val)


def STATS_SAVE(name, val):
    if fhe_stats:
        r = _stats.get(name)
        if r is None:
            r = _stats[name] = StatsRecord(name)
        r.save(val)


def print_stats(out=None):
    """src/fhe_stats.cpp:print_stats: name ave=... max=... (or the number of saved values)"""
    out = out or sys.stderr
    for name in sorted(_stats):
        r = _stats[name]
        if r.saved_values:
            out.write(f"{name} saved values: {len(r.saved_values)}\n")
        else:
            out.write(f"{name} ave={r.sum / r.count if r.count else 0.0} max={r.max}\n")


def fetch_saved_values(name):
    r = _stats.get(name)
    return r.saved_values if r is not None else None


def reset_stats():
    _stats.clear()

#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.x, sqlite `*_results.db`) kernel trace the way
`--stats` CSV used to: per-kernel calls / total / avg / min / max durations and share.
usage: python tools/rocpd_summary.py gpurun_out/prof1 > profiles/<name>.txt"""
import glob
import os
import sqlite3
import sys


def main():
    path = [a for a in sys.argv[1:] if not a.startswith("--")][0]
    dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True))
    for db in dbs:
        c = sqlite3.connect(db)
        cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
        rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                         "from kernels group by 1 order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print(f"# {os.path.basename(db)}  (durations in microseconds)")
        print(f"{'kernel':70s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
        for n, k, s, a, mn, mx in rows:
            short = n if len(n) <= 70 else n[:67] + "..."
            print(f"{short:70s} {k:6d} {s/1e3:12.1f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.2f}")
        if "--by-grid" in sys.argv and "grid_x" in cols:
            # the same kernel is launched at very different sizes (batch-1 key generation and
            # encryption next to the batch-128 timed loop): one line per (kernel, grid)
            print("# by launch size (workgroups = grid_x / workgroup_x); last20 = average of the last 20 launches of that")
            print("# size -- for ntt_row_kernel<14,fwd> at 6400 workgroups that is bench.py's roofline loop, which runs last")
            for n, gx, wx, k, a, mn, mx in c.execute(
                    f"select {name_col}, grid_x, workgroup_x, count(*), avg(end-start), min(end-start), max(end-start) "
                    "from kernels group by 1, 2 having count(*) >= 2 order by 1, 2 desc").fetchall():
                if "ntt_" in n or "break_digits" in n or "keyswitch" in n or "tensor" in n or "rns_extend" in n:
                    short = n if len(n) <= 60 else n[:57] + "..."
                    tail = c.execute(f"select avg(d) from (select end-start as d from kernels where {name_col} = ? and "
                                     "grid_x = ? order by start desc limit 20)", (n, gx)).fetchone()[0]
                    print(f"  {short:60s} wgs {gx // max(wx, 1):7d} calls {k:5d} avg {a/1e3:9.2f} us  min {mn/1e3:9.2f}  "
                          f"max {mx/1e3:9.2f}  last20 {tail/1e3:9.2f}")
        if "--gaps" in sys.argv:
            # where the device sat idle: the largest gaps between the end of one kernel and the
            # start of the next (host-bound stretches, synchronous API calls, copy engines)
            seq = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
            gaps = []
            for (n0, s0, e0), (n1, s1, e1) in zip(seq, seq[1:]):
                if s1 - e0 > 100000:
                    gaps.append((s1 - e0, n0, n1, e0 - seq[0][1]))
            print(f"# idle gaps > 100 us: {len(gaps)}, total {sum(g[0] for g in gaps)/1e6:.1f} ms; "
                  f"trace span {(seq[-1][2]-seq[0][1])/1e6:.1f} ms, busy {tot/1e6:.1f} ms")
            last = [g for g in gaps if g[3] > 0.5 * (seq[-1][2] - seq[0][1])]   # second half: the timed loops
            for g, n0, n1, at in sorted(last, reverse=True)[:25]:
                print(f"  {g/1e3:10.1f} us at +{at/1e6:9.1f} ms  after {n0[:48]:48s} before {n1[:48]}")
        try:
            pm = c.execute("select * from pmc_events limit 1").fetchall()
            if pm:
                pc = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
                print("# pmc_events columns:", pc)
        except Exception:
            pass


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel, per-launch-size PMC totals from rocprofv3 (ROCm 7.x sqlite) --pmc runs.
usage: python tools/rocpd_pmc.py gpurun_out/pmc1 [gpurun_out/pmc2 ...] [--min-us 20]
One line per (kernel, workgroups per launch, counter): dispatches, average duration, average counter value
per dispatch (summed over the counter's instances).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts a 128-byte request as 64 bytes (MI355X_MICROARCH.md): the `bytes` column doubles it."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    min_us = 20.0
    if "--min-us" in sys.argv:
        min_us = float(sys.argv[sys.argv.index("--min-us") + 1])
    for path in args:
        for db in sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True)):
            c = sqlite3.connect(db)
            rows = c.execute(
                "select p.name, k.grid_x * k.grid_y * k.grid_z / (k.workgroup_x * k.workgroup_y * k.workgroup_z), "
                "p.counter_name, p.dispatch_id, sum(p.counter_value), max(p.duration) "
                "from pmc_events p join kernels k on k.dispatch_id = p.dispatch_id "
                "group by p.name, p.counter_name, p.dispatch_id").fetchall()
            agg = defaultdict(list)
            for name, wgs, cn, did, val, d in rows:
                agg[(name, wgs, cn)].append((val, d))
            print(f"# {db}")
            print(f"{'kernel':64s} {'workgroups':>10s} {'counter':>14s} {'dispatches':>10s} {'avg_us':>9s} {'avg/dispatch':>16s} {'bytes':>16s}")
            for (name, wgs, cn), v in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
                avg_us = sum(x[1] for x in v) / len(v) / 1e3
                if avg_us < min_us:
                    continue
                avg = sum(x[0] for x in v) / len(v)
                byts = ""
                if cn == "FETCH_SIZE":
                    byts = f"{avg * 1024 * 2:.0f}"
                elif cn == "WRITE_SIZE":
                    byts = f"{avg * 1024:.0f}"
                short = name if len(name) <= 64 else name[:61] + "..."
                print(f"{short:64s} {wgs:10d} {cn:>14s} {len(v):10d} {avg_us:9.1f} {avg:16.1f} {byts:>16s}")


if __name__ == "__main__":
    main()

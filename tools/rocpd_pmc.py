#!/usr/bin/env python3
"""Per-kernel PMC totals from rocprofv3 (ROCm 7.x sqlite) --pmc runs.
usage: python tools/rocpd_pmc.py gpurun_out/pmc1 [gpurun_out/pmc2 ...]"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main():
    for path in sys.argv[1:]:
        for db in sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True)):
            c = sqlite3.connect(db)
            rows = c.execute("select name, counter_name, dispatch_id, sum(counter_value), max(duration) "
                             "from pmc_events group by name, counter_name, dispatch_id").fetchall()
            agg = defaultdict(lambda: defaultdict(list))
            dur = defaultdict(dict)
            for name, cn, did, val, d in rows:
                agg[name][cn].append(val)
                dur[name][did] = d
            print(f"# {db}")
            for name in sorted(agg, key=lambda n: -sum(dur[n].values())):
                nd = len(dur[name])
                print(f"{name[:100]}\n    dispatches={nd} avg_us={sum(dur[name].values())/nd/1e3:.1f}")
                for cn in sorted(agg[name]):
                    v = agg[name][cn]
                    print(f"    {cn:28s} avg/dispatch = {sum(v)/len(v):16.1f}")
                    if len(set(round(x) for x in v)) > 1 and len(v) <= 16:
                        # the same kernel at different launch sizes: one value per dispatch, in order
                        print(f"    {'':28s} per dispatch  = " + "  ".join(f"{x:.1f}" for x in v))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Kernel sequence of the last multiply of a tools/bench_levels.py run under rocprofv3
--kernel-trace (the level-2 loop runs last): everything between the last two break_digits launches.
usage: python tools/level2_sequence.py gpurun_out/<dir>"""
import glob
import os
import re
import sqlite3
import sys

db = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))[0]
c = sqlite3.connect(db)
rows = c.execute("select name, grid_x/workgroup_x, start, end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "break_digits" in r[0]]
a, b = idx[-2], idx[-1]
tot = 0.0
for r in rows[a + 1:b + 1]:
    n = re.sub(r"\(.*", "", r[0]).replace("void hx::", "").replace("hx::", "")
    d = (r[3] - r[2]) / 1e3
    tot += d
    print(f"{n[:52]:52s} wgs {r[1]:6d} {d:9.1f} us")
print(f"sum of kernel durations {tot:.1f} us, wall {(rows[b][3] - rows[a][3]) / 1e3:.1f} us")

#!/usr/bin/env python3
"""Print the kernel timeline of the last frame in a rocprofv3 (rocpd sqlite) kernel trace:
    python tools/wf_timeline.py <results.db> [name of the frame's first kernel, default pwf_init]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = sys.argv[2] if len(sys.argv) > 2 else "pwf_init"
rows = db.execute("select name, start, end, grid_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if first in r[0]]
last = rows[idx[-1]:]
t0 = last[0][1]
for r in last:
    print(f"{r[0][:44]:44s} start {(r[1] - t0) / 1e3:9.1f} us  dur {(r[2] - r[1]) / 1e3:9.1f} us  grid {r[3]}")
print("span us", (last[-1][2] - t0) / 1e3)

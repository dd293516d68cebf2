#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 rocpd database (the default output format):
   python experiments/rocpd_kernels.py <results.db> [name substring ...]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = db.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
by = defaultdict(list)
for name, s, e in rows:
    m = re.search(r"(\w+_kernel)", name)
    short = m.group(1) + ("<" + name.split("<", 1)[1].split(">")[0] + ">" if m and "<" in name else "") if m else name[:60]
    short = short[:60]
    by[short].append((e - s) / 1000.0)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(sys.argv) > 2 and not any(a in k for a in sys.argv[2:]):
        continue
    v2 = sorted(v)
    print("%-62s n=%5d  avg %9.2f us  med %9.2f  min %9.2f" % (k, len(v), sum(v) / len(v), v2[len(v) // 2], v2[0]))

#!/usr/bin/env python3
"""Kernels and memory copies of a rocprofv3 rocpd database on one time axis (the last `n` events):
   python experiments/rocpd_timeline.py <results.db> [n]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
ev = []
kc = [r[1] for r in db.execute("pragma table_info(kernels)")]
for name, s, e in db.execute("select name, start, end from kernels"):
    m = re.search(r"(\w+_kernel)", name)
    ev.append((s, e, "K " + (m.group(1) if m else name[:40])))
mc = [t for t in tabs if t in ("memory_copies", "memory_copy")]
if mc:
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % mc[0])]
    name_col = "name" if "name" in cols else cols[0]
    size_col = "size" if "size" in cols else None
    q = "select %s, start, end%s from %s" % (name_col, (", " + size_col) if size_col else "", mc[0])
    for row in db.execute(q):
        ev.append((row[1], row[2], "C %s %s" % (str(row[0])[:28], ("%.1f MB" % (row[3] / 1e6)) if size_col else "")))
else:
    print("no memory copy table:", [t for t in tabs if "copy" in t.lower()])
ev.sort()
ev = ev[-n:]
t0 = ev[0][0]
for s, e, what in ev:
    print("%10.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, what))

"""Idle time between consecutive kernels of a rocprofv3 kernel trace (rocpd SQLite): mean gap by (previous kernel -> next kernel).
  python tools/rocpd_gaps.py <results.db> [min_count]"""
import collections
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("ccalib::", "")
    return re.sub(r"[<(].*$", "", name)


db = sqlite3.connect(sys.argv[1])
min_count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = db.execute("select name, start, end from kernels order by start").fetchall()
gaps = collections.defaultdict(list)
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    gaps[(short(n0), short(n1))].append((s1 - e0) / 1e3)
for (a, b), g in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(g) < min_count:
        continue
    g.sort()
    print(f"{a:28s} -> {b:28s} n={len(g):5d} median gap {g[len(g) // 2]:8.2f} us   p10 {g[len(g) // 10]:8.2f}  p90 {g[len(g) * 9 // 10]:8.2f}")

"""Per-kernel launch durations grouped by grid size: python tools/rocpd_bygrid.py <results.db> <kernel substring>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
for name, grid, n, mean, mn, mx in db.execute(
        "select name, grid_x, count(*), avg(duration), min(duration), max(duration) from kernels where name like ? "
        "group by name, grid_x order by name, grid_x", (f"%{sys.argv[2]}%",)):
    print(f"{name[:50]:50s} grid={grid:8d} n={n:4d} mean={mean / 1e3:8.2f} min={mn / 1e3:8.2f} max={mx / 1e3:8.2f} us")

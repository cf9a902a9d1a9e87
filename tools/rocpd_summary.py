"""Summarises rocprofv3's rocpd SQLite output into the small CSV / JSON files kept under profiles/.

  python tools/rocpd_summary.py kernels <results.db> <out.csv>     per-kernel calls / total / mean / min / max
  python tools/rocpd_summary.py pmc <results.db> [<results.db>...] <out.json>   per-kernel mean of each counter
"""
import csv
import json
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("ccalib::", "")
    return re.sub(r"\(.*$", "", name)


def kernels(dbpath, out):
    db = sqlite3.connect(dbpath)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                      "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "mean_us", "min_us", "max_us", "pct", "vgpr", "agpr", "sgpr",
                    "lds_bytes", "scratch_bytes", "grid_threads", "block"])
        for r in rows:
            w.writerow([short(r[0]), r[1], f"{r[2] / 1e3:.3f}", f"{r[3] / 1e3:.3f}", f"{r[4] / 1e3:.3f}",
                        f"{r[5] / 1e3:.3f}", f"{100 * r[2] / total:.2f}", *r[6:]])
    for r in rows:
        print(f"{short(r[0]):60s} n={r[1]:4d} mean={r[3] / 1e3:9.3f} us  {100 * r[2] / total:5.1f}%  vgpr={r[6]} agpr={r[7]} "
              f"lds={r[9]} scratch={r[10]}")


def pmc(dbs, out):
    res = {}
    for p in dbs:
        db = sqlite3.connect(p)
        for name, counter, mean, n in db.execute("select kernel_name, counter_name, avg(value), count(*) from "
                                                 "counters_collection group by kernel_name, counter_name"):
            res.setdefault(short(name), {})[counter] = {"mean_per_launch": mean, "launches": n}
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        print(k, {c: round(x["mean_per_launch"], 3) for c, x in v.items()})


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "kernels":
        kernels(sys.argv[2], sys.argv[3])
    elif len(sys.argv) >= 4 and sys.argv[1] == "pmc" and not sys.argv[-1].endswith(".db"):
        pmc(sys.argv[2:-1], sys.argv[-1])
    else:
        sys.exit(__doc__)

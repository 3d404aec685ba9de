#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite output (ROCm 7.2 default) into the small CSV summaries kept under profiles/.

  python profiles/summarize_rocprof.py stats  gpurun_out/prof_stats/r01_results.db  profiles/r01_kernel_stats.csv
  python profiles/summarize_rocprof.py pmc    gpurun_out/prof_fetch/r01_results.db  profiles/r01_pmc_fetch.csv
"""
import csv
import sqlite3
import sys


def short(name):
    name = name.split("(")[0].replace("rgbl::", "")
    if name.startswith("void "):
        name = name[5:]
    if name.startswith("k_"):  # our kernels: drop the template arguments (k_fast_cells<48> -> k_fast_cells)
        name = name.split("<")[0]
    return name


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, total, avg, pct in rows:
            wr.writerow([short(name), calls, "%.3f" % total, "%.3f" % avg, "%.3f" % pct])


def pmc(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name order by avg(value) desc").fetchall()
    with open(out, "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel", "counter", "dispatches", "avg_value_per_dispatch", "avg_duration_ns"])
        for name, counter, n, val, dur in rows:
            wr.writerow([short(name), counter, n, "%.3f" % val, "%.1f" % (dur or 0)])


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])

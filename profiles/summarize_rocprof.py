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


def traffic(out_dir, steps, warmup, frames_per_step, out, tag="bench", extra_args=""):
    """FETCH_SIZE / WRITE_SIZE (and SQ_INSTS_VALU) passes of `bench.py --serial` (prof_<tag>_*) + of tools/micro/hbm_calib
    (prof_calib_*) -> bytes (vector instructions) per STEP per kernel (sum over the dispatches of the timed + profiled steps /
    their number), the bytes calibrated."""
    import glob
    import json
    import os

    def rows(tag):
        db = glob.glob(os.path.join(out_dir, "prof_%s" % tag, "**", "*.db"), recursive=True)
        if not db:
            return []
        cur = sqlite3.connect(db[0]).cursor()
        return cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()

    calib_bytes = float(1 << 30)
    factors = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for name, counter, n, total in rows("calib_" + c):
            k = short(name)
            if k.startswith("calib_") and total:
                factors["%s:%s" % (c, k)] = calib_bytes * n / total      # bytes per counter unit
    # the bench kernels move 4-byte (pixel kernels) to 16-byte (descriptor / cloud) pieces per lane: use the 4-byte factors,
    # and report both so that the choice can be checked
    f_fetch = factors.get("FETCH_SIZE:calib_read_b32")
    f_write = factors.get("WRITE_SIZE:calib_write_b32")
    # bench.py runs `warmup` + `steps` timed steps and bench.PROF_STEPS more for its per-kernel timing leg, all serialised
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    n_steps = int(steps) + int(warmup) + bench.PROF_STEPS
    kernels = {}
    for c, f, key in (("FETCH_SIZE", f_fetch, "fetch_bytes_per_step"), ("WRITE_SIZE", f_write, "write_bytes_per_step"),
                      ("SQ_INSTS_VALU", 1.0, "valu_insts_per_step"), ("SQ_ACTIVE_INST_VALU", 1.0, "valu_active_quadcycles_per_step")):
        for name, counter, n, total in rows(tag + "_" + c):
            k = short(name)
            if not k.startswith("k_"):
                continue
            d = kernels.setdefault(k, {"fetch_bytes_per_step": 0.0, "write_bytes_per_step": 0.0})
            d[key] = (total or 0.0) * (f or 0.0) / n_steps
            d["dispatches_per_step"] = n / n_steps
    json.dump({"frames_per_step": int(frames_per_step), "steps_profiled": n_steps,
               "bytes_per_counter_unit": factors, "factor_used": {"FETCH_SIZE": f_fetch, "WRITE_SIZE": f_write},
               "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE|SQ_INSTS_VALU -- python bench.py --serial --no-cpu-baseline --no-extras --steps %s --warmup %s %s" % (steps, warmup, extra_args),
               "kernels": kernels}, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:9])
    else:
        {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])

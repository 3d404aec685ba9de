#!/bin/bash
# Kernel timeline of the (overlapped, non-serial) bench: rocprofv3 --kernel-trace, dispatches of the last timed steps as CSV
# (name, queue, start_us, end_us relative to the first) in gpurun_out/timeline.csv.  Usage: gpurun -- 'bash tools/gpu_timeline.sh [bench args]'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $OUT/prof_tl -o tl -- python $REPO/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 "$@" > $OUT/prof_tl.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3, csv
db = glob.glob("gpurun_out/prof_tl/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = [t for t in tabs if t == "kernels"] or [t for t in tabs if "kernel" in t.lower()]
print("tables:", [t for t in tabs if "kernel" in t.lower()][:10])
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view[0])]
print(view[0], cols)
rows = cur.execute("select * from %s" % view[0]).fetchall()
ix = {c: i for i, c in enumerate(cols)}
name_c = "name" if "name" in ix else [c for c in cols if "name" in c][0]
st = "start" if "start" in ix else [c for c in cols if "start" in c][0]
en = "end" if "end" in ix else [c for c in cols if c.startswith("end")][0]
q = [c for c in ("stream_id", "queue_id") if c in ix]   # the HIP stream AND the HSA queue it was mapped to (streams can share one)
rows = [r for r in rows if "k_" in str(r[ix[name_c]])]
rows.sort(key=lambda r: r[ix[st]])
t0 = rows[0][ix[st]]
with open("gpurun_out/timeline.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "stream", "queue", "start_us", "end_us"])
    for r in rows:
        n = str(r[ix[name_c]]).split("(")[0].replace("rgbl::", "").replace("void ", "").split("<")[0]
        w.writerow([n, r[ix["stream_id"]] if "stream_id" in ix else "", r[ix["queue_id"]] if "queue_id" in ix else "",
                    "%.1f" % ((r[ix[st]] - t0) / 1e3), "%.1f" % ((r[ix[en]] - t0) / 1e3)])
print(len(rows), "dispatches")
PY
rm -rf $OUT/prof_tl

#!/bin/bash
# Kernel timeline of the single-frame host path (tools/host_api_latency.py): rocprofv3 --kernel-trace, the dispatches of one frame
# in the middle of the run, times relative to the frame's first kernel.  gpurun -- 'bash tools/gpu_timeline_single.sh'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD
cd /tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d $OUT/prof_tls -o tl -- python $REPO/tools/host_api_latency.py > $OUT/prof_tls.log 2>&1
cd $REPO
python - <<'PY'
import glob, sqlite3
db = glob.glob("gpurun_out/prof_tls/**/*.db", recursive=True)[0]
con = sqlite3.connect(db); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
ix = {c: i for i, c in enumerate(cols)}
rows = [(r[ix["start"]], r[ix["end"]], str(r[ix["name"]]).split("(")[0].replace("rgbl::", "").replace("void ", "").split("<")[0], r[ix["queue_id"]] if "queue_id" in ix else 0) for r in cur.execute("select * from kernels")]
try:
    mc = [r[1] for r in cur.execute("pragma table_info(memory_copies)")]
    mi = {c: i for i, c in enumerate(mc)}
    rows += [(r[mi["start"]], r[mi["end"]], "copy:" + str(r[mi["name"]])[-12:] + ":%d" % r[mi["size"]], -1) for r in cur.execute("select * from memory_copies")]
except Exception as e:
    print("no memory copies:", e)
rows.sort()
# frames: a frame starts with the first k_fast_cells / k_resize after a hamming kernel
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_hamming_fp4")]
mid = starts[len(starts) // 2]
seg = rows[mid + 1: starts[len(starts) // 2 + 1] + 3]
t0 = seg[0][0]
for s, e, n, q in seg:
    print("%-28s q=%-3s %7.1f -> %7.1f (%5.1f)" % (n[:28], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
PY
rm -rf $OUT/prof_tls

#!/bin/bash
# Per-call latency of the matcher entry points (bench_calls.py) + the rocprofv3 kernel statistics of the same command.
#   Usage: gpurun --timeout 900 -- 'bash tools/gpu_calls.sh [tag]'   -> gpurun_out/<tag>_tracking_calls.json, <tag>_tracking_calls_kernel_stats.csv
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-r06}
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; REPO=$PWD
timeout 400 python bench_calls.py > $OUT/${TAG}_tracking_calls.json 2> $OUT/${TAG}_tracking_calls.err
python - "$OUT/${TAG}_tracking_calls.json" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for leg, v in d.items():
        for k, e in v.items():
            if isinstance(e, dict):
                print(leg, k, "gpu", e["gpu_call"], "kernels", e.get("kernels_total_us"), "cpu", e.get("cpu_reference"), e["parity"][:20])
except Exception as ex:
    print("calls failed", ex, open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
cd /tmp
RGBL_CALL_BENCH_N=50 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_calls -o calls -- python $REPO/bench_calls.py > $OUT/prof_calls.log 2>&1
cd $REPO
db=$(find $OUT/prof_calls -name "*.db" | head -1); [ -n "$db" ] && python profiles/summarize_rocprof.py stats $db $OUT/${TAG}_tracking_calls_kernel_stats.csv
rm -rf $OUT/prof_calls
head -20 $OUT/${TAG}_tracking_calls_kernel_stats.csv

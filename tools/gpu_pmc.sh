#!/bin/bash
# One rocprofv3 counter pass over the serialised bench: bash tools/gpu_pmc.sh <tag> "<counters>" [ENV=...]...
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD
tag=$1; counters=$2; shift 2
cd /tmp
env "$@" timeout 120 rocprofv3 --pmc $counters -d $OUT/prof_$tag -o p -- python $REPO/bench.py --serial --no-cpu-baseline --no-extras --steps 3 --warmup 1 > $OUT/prof_$tag.log 2>&1
cd $REPO
db=$(find $OUT/prof_$tag -name "*.db" | head -1)
[ -n "$db" ] && python profiles/summarize_rocprof.py pmc $db $OUT/pmc_$tag.csv
rm -rf $OUT/prof_$tag; tail -2 $OUT/prof_$tag.log
python - $OUT/pmc_$tag.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(dict)
for r in rows:
    if r["kernel"].startswith("k_"):
        by[r["kernel"]][r["counter"]] = (float(r["avg_value_per_dispatch"]), int(r["dispatches"]), float(r["avg_duration_ns"]))
for k, c in sorted(by.items()):
    d = next(iter(c.values()))
    print("%-20s n=%-4d %8.1f us  " % (k, d[1], d[2] / 1e3) + "  ".join("%s=%.4g" % (n.replace("SQ_", ""), v[0]) for n, v in sorted(c.items())))
PY

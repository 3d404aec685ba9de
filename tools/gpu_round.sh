#!/bin/bash
# One gpurun call: GPU parity tests, the bench line, rocprofv3 kernel stats and the two HBM counter
# passes.  Everything lands in gpurun_out/.  Usage: gpurun --timeout 1080 -- 'bash tools/gpu_round.sh'
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
( time timeout 540 python -m pytest tests -m gpu -q -n 4 ) > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
timeout 200 python bench.py > $OUT/bench.json 2> $OUT/bench.err


python - <<'PY'
import json
for n in ("bench",):
    try:
        d = json.loads(open("gpurun_out/%s.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["value"]), d["parity_spot_check"], {k: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_step"].items()})
    except Exception as e:
        print(n, "failed", e)
PY
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o r01 -- python $OLDPWD/bench.py --serial --no-cpu-baseline --steps 10 --warmup 3 > $OUT/prof_stats.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE -d $OUT/prof_fetch -o r01 -- python $OLDPWD/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/prof_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE -d $OUT/prof_write -o r01 -- python $OLDPWD/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/prof_write.log 2>&1
cd $OLDPWD
for k in stats fetch write; do
  db=$(find $OUT/prof_$k -name "*.db" | head -1)
  [ -n "$db" ] && python profiles/summarize_rocprof.py $([ $k = stats ] && echo stats || echo pmc) $db $OUT/r01_$k.csv
done
rm -rf $OUT/prof_stats $OUT/prof_fetch $OUT/prof_write   # the databases are large; the CSV summaries are what is kept
ls -la $OUT

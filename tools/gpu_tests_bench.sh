#!/bin/bash
# GPU parity tests + the default bench line, one gpurun call.  Usage: gpurun --timeout 900 -- 'bash tools/gpu_tests_bench.sh'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$PWD/gpurun_out
( time timeout 600 python -m pytest tests -m gpu -q -n 4 -x ) > $OUT/tests.log 2>&1
tail -15 $OUT/tests.log
timeout 400 python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("bench", round(d["value"]), d["parity_spot_check"][:40], {k: round(v, 3) for k, v in r["kernels_ms_per_step"].items()})
    print("roofline", r["bound"], r["kernel"], r["frac"], r.get("valu_issue_frac"), r.get("traffic_over_algorithmic"))
    print("cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"])
    for k, v in d.get("extra", {}).items():
        print(k, {kk: (vv if not isinstance(vv, dict) else {a: b for a, b in vv.items() if a in ("bound", "kernel", "frac", "total", "extract", "kernels_ms_per_step")}) for kk, vv in v.items() if kk != "what"})
except Exception as e:
    print("bench failed", e, open("gpurun_out/bench.err").read()[-1500:])
PY

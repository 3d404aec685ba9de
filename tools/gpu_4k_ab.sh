#!/bin/bash
# A/B of environment settings on the 4K workload (and the KITTI one) inside one gpurun call: every argument is "ENV=.. ENV=.. -- bench args"
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for setting in "$@"; do
  envs="${setting%%--*}"; args="--${setting#*--}"
  env $envs timeout 280 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 $args > gpurun_out/b.json 2>gpurun_out/b.err
  python - "$setting" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
    print("%-44s %7d frames/s %7.3f ms %s  %s" % (sys.argv[1], round(d["value"]), d["ms_per_step"], d["parity_spot_check"][:9],
          " ".join("%s=%.3f" % (k[2:], v) for k, v in d["roofline"]["kernels_ms_per_step"].items())))
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/b.err").read()[-600:])
PY
done

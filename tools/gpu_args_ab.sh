#!/bin/bash
# A/B of bench.py argument / environment sets inside one gpurun call: every argument is "ENV=.. ENV=.. -- bench args" ("-" = defaults)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for setting in "$@"; do
  envs=""; args=""
  if [ "$setting" != "-" ]; then envs="${setting%%--*}"; [ "$setting" != "${setting#*--}" ] && args="--${setting#*--}"; fi
  env $envs python bench.py --no-cpu-baseline --no-extras $args > gpurun_out/b.json 2>gpurun_out/b.err
  python - "$setting" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
    print("%-34s %7d frames/s %6.3f ms  %s" % (sys.argv[1], round(d["value"]), d["ms_per_step"], d["parity_spot_check"][:9]))
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/b.err").read()[-600:])
PY
done

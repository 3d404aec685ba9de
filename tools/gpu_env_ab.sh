#!/bin/bash
# A/B of environment settings on the bench: bash tools/gpu_env_ab.sh "A=1" "A=2 B=3" ...   ("-" = no setting)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for setting in "$@"; do
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs python bench.py --no-cpu-baseline --no-extras > gpurun_out/b.json 2>gpurun_out/b.err
  python - "$setting" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
    print("%-28s %7d frames/s %6.3f ms  %s  %s" % (sys.argv[1], round(d["value"]), d["ms_per_step"], d["parity_spot_check"][:9], {k[2:]: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_step"].items()}))
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/b.err").read()[-400:])
PY
done

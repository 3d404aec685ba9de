#!/bin/bash
# A/B of environment-selected kernel variants inside one gpurun call (same box): every argument is one environment
# setting ("-" = defaults), e.g.  gpurun --timeout 300 -- 'bash tools/gpu_ab.sh - RGBL_GAUSS_DOT=0'
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for setting in "$@"; do
  tag=$(echo "$setting" | tr -c 'A-Za-z0-9_=\n' '_')
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  env $envs timeout 150 python bench.py --no-cpu-baseline --no-extras > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ab_%s.json" % tag).read().strip().splitlines()[-1])
    print("%-28s %7.0f frames/s  %s  %s" % (tag, d["value"], "exact" if d["parity_spot_check"].startswith("bit-exact") else "PARITY?",
          " ".join("%s=%.3f" % (k[2:], v) for k, v in d["roofline"]["kernels_ms_per_step"].items())))
except Exception as e:
    print(tag, "failed", e, open("gpurun_out/ab_%s.err" % tag).read()[-600:])
PY
done

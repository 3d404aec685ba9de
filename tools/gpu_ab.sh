#!/bin/bash
# ONE A/B script for a gpurun call (same box for every setting).  Every argument is a setting
#     [lib.so|-][,ENV=..,ENV=..][ -- bench.py args]
#   lib.so   a library build (under exp/, which travels with the snapshot) that takes the product library's place; "-" or nothing
#            = the library as built
#   ENV=..   environment switches (include/rgbl_frontend.h lists them), comma separated
#   -- args  bench.py arguments (e.g. "--workload 4k --steps 10")
# A setting that starts with "sf:" measures the single-frame latency of the C++ drop-in classes (tools/shim_latency.py) instead
# of the batch step.  Examples:
#   gpurun --timeout 900 -- 'bash tools/gpu_ab.sh - ,RGBL_CONE=0 ",RGBL_COMPACT=0 -- --workload 4k" exp/lib_prev.so sf:,RGBL_LEVEL_SPLIT=0'
# (replaces the round 2 - 5 one-off scripts gpu_{4k,args,compact,env,lib,lib_env,octree_hist,sf_env,single_frame}_ab.sh)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
cp orb_slam3_rgbl_amd/librgbl_frontend.so /tmp/lib_asbuilt.so
for setting in "$@"; do
  spec="$setting"; sf=0
  case "$spec" in sf:*) sf=1; spec="${spec#sf:}";; esac
  args=""; case "$spec" in *" -- "*) args="${spec#* -- }"; spec="${spec%% -- *}";; esac
  lib="${spec%%,*}"; envs=""; [ "$spec" != "$lib" ] && envs=$(echo "${spec#*,}" | tr ',' ' ')
  if [ -z "$lib" ] || [ "$lib" = "-" ]; then cp /tmp/lib_asbuilt.so orb_slam3_rgbl_amd/librgbl_frontend.so; else cp "$lib" orb_slam3_rgbl_amd/librgbl_frontend.so; fi
  if [ $sf = 1 ]; then
    echo "== $setting"; env $envs python tools/shim_latency.py 2>&1 | grep "ms per frame"
    continue
  fi
  env $envs timeout 280 python bench.py --no-cpu-baseline --no-extras $args > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$setting" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
    print("%-40s %7d frames/s %7.3f ms %s  %s" % (sys.argv[1], round(d["value"]), d["ms_per_step"], "exact" if d["parity_spot_check"].startswith("bit-exact") else "PARITY?",
          " ".join("%s=%.3f" % (k[2:], v) for k, v in d["roofline"]["kernels_ms_per_step"].items())))
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/ab.err").read()[-600:])
PY
done
cp /tmp/lib_asbuilt.so orb_slam3_rgbl_amd/librgbl_frontend.so

#!/bin/bash
# A/B of library builds and environment settings inside one gpurun call (same box).  Every argument is "lib[,ENV=..,ENV=..]";
# lib = a .so under exp/ (travels with the snapshot) or "-" for the library as built.  The per-kernel times are the serialised leg's.
#   gpurun --timeout 600 -- 'bash tools/gpu_lib_env_ab.sh exp/lib_prev.so - -,RGBL_FAST_BS=64'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
cp orb_slam3_rgbl_amd/librgbl_frontend.so /tmp/lib_asbuilt.so
for spec in "$@"; do
  lib="${spec%%,*}"; envs=""; [ "$spec" != "$lib" ] && envs=$(echo "${spec#*,}" | tr ',' ' ')
  if [ "$lib" = "-" ]; then cp /tmp/lib_asbuilt.so orb_slam3_rgbl_amd/librgbl_frontend.so; else cp "$lib" orb_slam3_rgbl_amd/librgbl_frontend.so; fi
  env $envs timeout 200 python bench.py --no-cpu-baseline --no-extras $BENCH_ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - "$spec" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
    print("%-28s %7.0f frames/s %6.3f ms %s  %s" % (tag, d["value"], d["ms_per_step"], "exact" if d["parity_spot_check"].startswith("bit-exact") else "PARITY?",
          " ".join("%s=%.3f" % (k[2:], v) for k, v in d["roofline"]["kernels_ms_per_step"].items())))
except Exception as e:
    print(tag, "failed", e, open("gpurun_out/ab.err").read()[-600:])
PY
done
cp /tmp/lib_asbuilt.so orb_slam3_rgbl_amd/librgbl_frontend.so

#!/bin/bash
# Single-frame latency A/B of environment settings (C++ drop-in classes, tools/shim_latency.cpp), each setting twice:
#   gpurun -- 'bash tools/gpu_sf_env_ab.sh - RGBL_SF_ORDER=1'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for setting in "$@"; do
  if [ "$setting" = "-" ]; then envs=""; else envs="$setting"; fi
  echo "== $setting"; env $envs python tools/shim_latency.py 2>&1 | grep "ms per frame"
done
done

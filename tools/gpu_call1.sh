#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
( time timeout 400 python -m pytest tests -m gpu -q -n 4 -x ) > gpurun_out/tests.log 2>&1
tail -5 gpurun_out/tests.log
bash tools/gpu_ab.sh - RGBL_BF_MFMA=0

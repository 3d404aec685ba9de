#!/bin/bash
# Round 4: the strip kernel - parity on the GPU, then A/B against the cell-per-workgroup kernel (RGBL_FAST_STRIP=0) on both workloads
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -n 4 -k "fast_strip or waves_per_cell or cfg2 or 4k" ) > gpurun_out/tests_strip.log 2>&1; tail -3 gpurun_out/tests_strip.log
bash tools/gpu_4k_ab.sh "RGBL_FAST_STRIP=0 --workload kitti --steps 30" "RGBL_FAST_STRIP=8 --workload kitti --steps 30" "RGBL_FAST_STRIP=4 --workload kitti --steps 30" "RGBL_FAST_STRIP=16 --workload kitti --steps 30" "RGBL_FAST_STRIP=0 --workload 4k" "RGBL_FAST_STRIP=8 --workload 4k" "RGBL_FAST_STRIP=16 --workload 4k" "RGBL_FAST_STRIP=32 --workload 4k"

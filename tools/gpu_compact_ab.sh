#!/bin/bash
# Round 4: k_compact_cells - parity on the GPU, then A/B against the per-cell reservation (RGBL_COMPACT=0) on both workloads
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -n 4 -k "compaction or waves_per_cell or cfg2 or 4k or batch or under_load" ) > gpurun_out/tests_compact.log 2>&1; tail -3 gpurun_out/tests_compact.log
bash tools/gpu_4k_ab.sh "RGBL_COMPACT=0 --workload kitti --steps 30" " --workload kitti --steps 30" "RGBL_COMPACT=0 --workload 4k" " --workload 4k"  "RGBL_COMPACT=0 --workload kitti --steps 30" " --workload kitti --steps 30"

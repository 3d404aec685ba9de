#!/bin/bash
# Round 4: the quad-tree's breadth-first phase on the cell pyramid (default) against the round-by-round key passes (RGBL_OCTREE_HIST=0)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_parity_gpu.py tests/test_natural_images.py tests/test_shim.py -m gpu -q -x -n 4 ) > gpurun_out/tests_hist.log 2>&1; tail -2 gpurun_out/tests_hist.log
for h in 0 1; do echo "== RGBL_OCTREE_HIST=$h"; RGBL_OCTREE_HIST=$h python tools/octree_stamps.py 1241 376 2000 1 | grep "level [01]"; RGBL_OCTREE_HIST=$h python tools/octree_stamps.py 3840 2160 8000 1 | grep "level [01]"; RGBL_OCTREE_HIST=$h python tools/shim_latency.py 2>&1 | grep "ms per frame"; done
bash tools/gpu_4k_ab.sh "RGBL_OCTREE_HIST=0 --workload kitti --steps 30" "RGBL_OCTREE_HIST=1 --workload kitti --steps 30" "RGBL_OCTREE_HIST=0 --workload 4k" "RGBL_OCTREE_HIST=1 --workload 4k" "RGBL_OCTREE_HIST=0 --workload kitti --steps 30" "RGBL_OCTREE_HIST=1 --workload kitti --steps 30"

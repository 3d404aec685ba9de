#!/bin/bash
# Round 4: single-frame latency - quad-tree with its candidate lists in LDS (RGBL_OCTREE_LDSKEYS) and the levels 1-2 on a third
# stream (RGBL_LEVEL_SPLIT), each switched off / on; C++ drop-in classes (tools/shim_latency.cpp) and the kernel trace of the final setting
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_shim.py tests/test_natural_images.py -m gpu -q -x -n 4 ) > gpurun_out/tests_sf.log 2>&1; tail -2 gpurun_out/tests_sf.log
for env in "RGBL_OCTREE_LDSKEYS=0 RGBL_LEVEL_SPLIT=0" "RGBL_OCTREE_LDSKEYS=1 RGBL_LEVEL_SPLIT=0" "RGBL_OCTREE_LDSKEYS=0 RGBL_LEVEL_SPLIT=3" "RGBL_OCTREE_LDSKEYS=1 RGBL_LEVEL_SPLIT=3" "RGBL_OCTREE_LDSKEYS=1 RGBL_LEVEL_SPLIT=2" "RGBL_OCTREE_LDSKEYS=1 RGBL_LEVEL_SPLIT=4"; do
  echo "== $env"; env $env python tools/shim_latency.py 2>&1 | grep "ms per frame"
done
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/prof_single -o r04 -- python $OLDPWD/tools/host_api_latency.py > $OLDPWD/gpurun_out/prof_single.log 2>&1
cd $OLDPWD
db=$(find gpurun_out/prof_single -name "*.db" | head -1); [ -n "$db" ] && python profiles/summarize_rocprof.py stats $db gpurun_out/r04_single_frame_kernel_stats.csv
rm -rf gpurun_out/prof_single; head -14 gpurun_out/r04_single_frame_kernel_stats.csv; tail -3 gpurun_out/prof_single.log

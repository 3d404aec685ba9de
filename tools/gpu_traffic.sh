#!/bin/bash
# HBM traffic of the bench kernels for the roofline object: FETCH_SIZE and WRITE_SIZE passes (separate, as the counters do not
# fit one pass) over `bench.py --serial` (dispatches per step = launches per step), and the same two passes over the
# known-byte-count kernels of tools/micro/hbm_calib.hip, which give the bytes-per-counter-unit factors.  Writes
# gpurun_out/r02_pmc_traffic.json (copy to profiles/).   Usage: gpurun --timeout 900 -- 'bash tools/gpu_traffic.sh'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD
STEPS=6; WARM=2
hipcc -O3 --offload-arch=gfx950 -o /tmp/hbm_calib tools/micro/hbm_calib.hip || exit 1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c -d $OUT/prof_calib_$c -o p -- /tmp/hbm_calib > $OUT/prof_calib_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c -d $OUT/prof_bench_$c -o p -- python $REPO/bench.py --serial --no-cpu-baseline --no-extras --steps $STEPS --warmup $WARM > $OUT/prof_bench_$c.log 2>&1
done
cd $REPO
BATCH=$(python -c "import bench; print(bench.WORKLOADS['kitti'][4])")
python profiles/summarize_rocprof.py traffic $OUT $STEPS $WARM $BATCH $OUT/r02_pmc_traffic.json
for c in FETCH_SIZE WRITE_SIZE; do rm -rf $OUT/prof_calib_$c $OUT/prof_bench_$c; done
cat $OUT/r02_pmc_traffic.json | head -60

#!/bin/bash
# HBM traffic and vector-instruction counts of the bench kernels for the roofline object: FETCH_SIZE, WRITE_SIZE and
# SQ_INSTS_VALU passes (separate, one counter each) over `bench.py --serial` (dispatches per step = launches per step) for the
# KITTI workload and for the 4K one, and the two byte passes over the known-byte-count kernels of tools/micro/hbm_calib.hip,
# which give the bytes-per-counter-unit factors.  Writes gpurun_out/r06_pmc_traffic.json and r06_pmc_traffic_4k.json (copy to
# profiles/).   Usage: gpurun --timeout 900 -- 'bash tools/gpu_traffic.sh'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$PWD/gpurun_out; REPO=$PWD
STEPS=6; WARM=2
hipcc -O3 --offload-arch=gfx950 -o /tmp/hbm_calib tools/micro/hbm_calib.hip || exit 1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c -d $OUT/prof_calib_$c -o p -- /tmp/hbm_calib > $OUT/prof_calib_$c.log 2>&1
done
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU; do
  timeout 150 rocprofv3 --pmc $c -d $OUT/prof_bench_$c -o p -- python $REPO/bench.py --serial --no-cpu-baseline --no-extras --steps $STEPS --warmup $WARM > $OUT/prof_bench_$c.log 2>&1
  timeout 150 rocprofv3 --pmc $c -d $OUT/prof_4k_$c -o p -- python $REPO/bench.py --workload 4k --serial --no-cpu-baseline --no-extras --steps $STEPS --warmup $WARM > $OUT/prof_4k_$c.log 2>&1
done
cd $REPO
python profiles/summarize_rocprof.py traffic $OUT $STEPS $WARM $(python -c "import bench; print(bench.WORKLOADS['kitti'][4])") $OUT/r06_pmc_traffic.json bench
python profiles/summarize_rocprof.py traffic $OUT $STEPS $WARM $(python -c "import bench; print(bench.WORKLOADS['4k'][4])") $OUT/r06_pmc_traffic_4k.json 4k "--workload 4k"
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU; do rm -rf $OUT/prof_calib_$c $OUT/prof_bench_$c $OUT/prof_4k_$c; done
python - <<'PY'
import json
for n in ("r06_pmc_traffic.json", "r06_pmc_traffic_4k.json"):
    d = json.load(open("gpurun_out/" + n))
    print(n, {k: {a: round(b / 1e6, 1) for a, b in v.items() if a != "dispatches_per_step"} for k, v in d["kernels"].items()})
PY

#!/bin/bash
# One gpurun call for the round's evidence: GPU parity tests, the default bench line, rocprofv3 kernel stats of the
# serialised bench and of the single-frame host path, the calibrated FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU passes
# (tools/gpu_traffic.sh) and the SQ counter passes.  Everything lands in gpurun_out/ (copy the r06_* files to profiles/).
#   Usage: gpurun --timeout 1700 -- 'bash tools/gpu_round.sh'
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; REPO=$PWD
( time timeout 540 python -m pytest tests -m gpu -q -n 4 ) > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
try:
    d = [json.loads(l) for l in open("gpurun_out/bench.json") if l.startswith("{")][-1]
    print("bench", round(d["value"]), d["parity_spot_check"][:30], {k: round(v, 3) for k, v in d["roofline"]["kernels_ms_per_step"].items()})
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("all_cores", {}).get("value"))
    print("extra", {k: {kk: vv for kk, vv in v.items() if kk not in ("what", "roofline")} for k, v in d.get("extra", {}).items()})
except Exception as e:
    print("bench failed", e, open("gpurun_out/bench.err").read()[-800:])
PY
cp $OUT/bench.json $OUT/r06_bench_line.json
cd /tmp
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o r06 -- python $REPO/bench.py --serial --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $OUT/prof_stats.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/prof_single -o r06 -- python $REPO/tools/host_api_latency.py > $OUT/prof_single.log 2>&1
cd $REPO
db=$(find $OUT/prof_stats -name "*.db" | head -1); [ -n "$db" ] && python profiles/summarize_rocprof.py stats $db $OUT/r06_kernel_stats.csv
db=$(find $OUT/prof_single -name "*.db" | head -1); [ -n "$db" ] && python profiles/summarize_rocprof.py stats $db $OUT/r06_single_frame_kernel_stats.csv
rm -rf $OUT/prof_stats $OUT/prof_single
bash tools/gpu_traffic.sh > $OUT/traffic.log 2>&1; tail -3 $OUT/traffic.log
bash tools/gpu_pmc.sh sqA "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" > $OUT/sqA.log 2>&1
bash tools/gpu_pmc.sh sqB "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" > $OUT/sqB.log 2>&1
cp $OUT/pmc_sqA.csv $OUT/r06_pmc_sq_cycles.csv; cp $OUT/pmc_sqB.csv $OUT/r06_pmc_sq_insts.csv
bash tools/gpu_timeline.sh > $OUT/timeline.log 2>&1; cp $OUT/timeline.csv $OUT/r06_step_timeline.csv
( cd tools/micro && ./valu_rates ) > $OUT/r06_valu_rates.txt 2>&1
cat $OUT/r06_kernel_stats.csv | head -14; cat $OUT/r06_single_frame_kernel_stats.csv | head -14; cat $OUT/r06_valu_rates.txt
bash tools/gpu_calls.sh r06 > $OUT/calls.log 2>&1; grep -E "^cfg3|^tracking" $OUT/calls.log

#!/bin/bash
# Round 4, VERDICT r3 item 3: why is k_fast_cells 63 % slower per pixel on 4K frames?  SQ counter passes on --workload 4k
# and an A/B over batch size and the XCD mapping, all inside one gpurun call.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
A="bench.py --workload 4k --serial --no-cpu-baseline --no-extras --steps 3 --warmup 1"
bash tools/gpu_pmc_cmd.sh 4k_sqA "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" $A
bash tools/gpu_pmc_cmd.sh 4k_sqB "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES" $A
bash tools/gpu_pmc_cmd.sh 4k_sqC "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE" $A
bash tools/gpu_pmc_cmd.sh k_sqC "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LEVEL_WAVES GRBM_GUI_ACTIVE" bench.py --serial --no-cpu-baseline --no-extras --steps 3 --warmup 1
for setting in "--workload 4k --batch 64" "--workload 4k --batch 256" "RGBL_XCD_MAP=0 --workload 4k --batch 64" "--workload 4k --batch 64 --serial"; do
  envs=""; args="$setting"
  case "$setting" in RGBL_*) envs="${setting%% --*}"; args="--${setting#* --}";; esac
  env $envs timeout 280 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 $args > gpurun_out/b.json 2>gpurun_out/b.err
  python - "$setting" <<'PY'
import json, sys
try:
    d = json.loads(open("gpurun_out/b.json").read().strip().splitlines()[-1])
    print("%-40s %7d frames/s %7.3f ms %s  %s" % (sys.argv[1], round(d["value"]), d["ms_per_step"], d["parity_spot_check"][:9],
          " ".join("%s=%.3f" % (k[2:], v) for k, v in d["roofline"]["kernels_ms_per_step"].items())))
except Exception as e:
    print(sys.argv[1], "failed", e, open("gpurun_out/b.err").read()[-600:])
PY
done

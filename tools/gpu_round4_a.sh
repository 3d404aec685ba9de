#!/bin/bash
# Round 4, first evidence call: the new GPU tests (C-ABI gather over RCCL, prefetch regression), then the bench line with the gather legs.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 500 python -m pytest tests/test_distributed.py tests/test_gather_cpp.py tests/test_parity_gpu.py -m gpu -q -x -n 4 -k "gather or rccl or partial or overlapped or pack" ) > gpurun_out/tests_a.log 2>&1
tail -5 gpurun_out/tests_a.log
timeout 500 python bench.py --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
python - <<'PY'
import json
try:
    d = [json.loads(l) for l in open("gpurun_out/bench_a.json") if l.startswith("{")][-1]
    print("bench", round(d["value"]), d["parity_spot_check"][:9], d["roofline"]["step_algorithmic_GB/s"], d["roofline"]["step_kernel_sum_GB/s"])
    print("gather legs", d["extra"]["gather_rccl_1rank"])
    print("4k", {k: v for k, v in d["extra"]["cfg5_4k"].items() if k not in ("what", "roofline")})
except Exception as e:
    print("bench failed", e, open("gpurun_out/bench_a.err").read()[-1500:])
PY

#!/bin/bash
# A/B of whole library builds inside one gpurun call (same box): every argument is a .so (built into exp/, which travels with
# the snapshot) that takes the product library's place for one bench run; "-" = the library as built.
#   gpurun --timeout 600 -- 'bash tools/gpu_lib_ab.sh exp/lib_head.so -'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
cp orb_slam3_rgbl_amd/librgbl_frontend.so /tmp/lib_asbuilt.so
for lib in "$@"; do
  if [ "$lib" = "-" ]; then cp /tmp/lib_asbuilt.so orb_slam3_rgbl_amd/librgbl_frontend.so; else cp "$lib" orb_slam3_rgbl_amd/librgbl_frontend.so; fi
  tag=$(basename "$lib" .so)
  timeout 200 python bench.py --no-cpu-baseline --no-extras $BENCH_ARGS > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ab_%s.json" % tag).read().strip().splitlines()[-1])
    print("%-16s %7.0f frames/s %6.3f ms %s  %s" % (tag, d["value"], d["ms_per_step"], "exact" if d["parity_spot_check"].startswith("bit-exact") else "PARITY?",
          " ".join("%s=%.3f" % (k[2:], v) for k, v in d["roofline"]["kernels_ms_per_step"].items())))
except Exception as e:
    print(tag, "failed", e, open("gpurun_out/ab_%s.err" % tag).read()[-600:])
PY
done
cp /tmp/lib_asbuilt.so orb_slam3_rgbl_amd/librgbl_frontend.so

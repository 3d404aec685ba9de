#!/bin/bash
# quick look at the quad-tree kernel: parity of the extractor tests, phase stamps of one frame, bench line per counting mode
python -m pytest tests -m gpu -x -q -k "extractor" 2>&1 | tail -2
python tools/octree_stamps.py 1241 376 2000 1 | grep "level 0"
for m in ${MODES:-0}; do
RGBL_OCTREE_MODE=$m python bench.py --no-cpu-baseline ${EXTRA:---no-extras} > gpurun_out/bench.json 2> gpurun_out/bench.err
python - $m <<'PY'
import json, sys
d=json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print('mode', sys.argv[1], round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['kernels_ms_per_step'].items()})
if 'extra' in d: print(d['extra']['host_api_single_frame']['ms_per_frame'], round(d['extra']['cfg5_4k']['frames_per_s']), round(d['extra']['cfg3_stereo']['stereo_frames_per_s']))
PY
done

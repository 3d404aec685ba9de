cd "${GRAFT_REPO_ROOT:-.}"
for b in 256 384 512 768 1024; do
  python bench.py --no-cpu-baseline --no-extras --batch $b --steps 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch', d['config']['frames_per_gpu_per_step'], round(d['value']), round(d['ms_per_step'],3))"
done
for q in 2 3 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-extras --steps 12 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues $q', round(d['value']), round(d['ms_per_step'],3))"
done

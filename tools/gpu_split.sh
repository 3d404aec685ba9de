cd "${GRAFT_REPO_ROOT:-.}"
run() { env $1 python bench.py --no-cpu-baseline --no-extras --steps 16 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', round(d['value']), round(d['ms_per_step'],3), d['parity_spot_check'][:9])"; }
run A=1 ""
run RGBL_MATCHER_STREAM=own ""
run A=1 ""
run RGBL_MATCHER_STREAM=own ""
run RGBL_MATCHER_STREAM=own "--split 2"

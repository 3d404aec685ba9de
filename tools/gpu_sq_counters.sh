cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$PWD/gpurun_out; cd /tmp
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/prof_sq -o r01 -- python $OLDPWD/bench.py --serial --no-cpu-baseline --steps 4 --warmup 2 > $OUT/prof_sq.log 2>&1
cd $OLDPWD
db=$(find $OUT/prof_sq -name "*.db" | head -1)
[ -n "$db" ] && python profiles/summarize_rocprof.py pmc $db $OUT/r01_sq.csv
rm -rf $OUT/prof_sq; tail -3 $OUT/prof_sq.log; head -30 $OUT/r01_sq.csv

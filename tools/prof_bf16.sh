# rocprofv3 kernel stats of a short bf16 bench run -> gpurun_out/prof_bf16/
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_bf16
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bf16 -o bf16 -- python3 $GRAFT_REPO_ROOT/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-graph --streams 1 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_bf16.log 2>&1
cd $GRAFT_REPO_ROOT && ls gpurun_out/prof_bf16 | head; f=$(ls gpurun_out/prof_bf16/*kernel_stats.csv | head -1); python3 - <<PY
import csv
rows=list(csv.DictReader(open('$f')))
for r in rows[:30]:
    print('%-90s calls %5s avg_us %9.1f  pct %5s' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY

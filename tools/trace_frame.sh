# usage: tools/trace_frame.sh <tag> [bench args] [-- ENV=VAL..]  -- kernel trace of a short bench run; prints one frame's timeline
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$tag -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 6 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/trace_$tag.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/trace_$tag/*kernel_trace.csv gpurun_out/trace_$tag/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/frame_timeline.py $f -4 > gpurun_out/timeline_$tag.txt
python3 - $f <<'PY' >> gpurun_out/timeline_$tag.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 60% of the run: union of busy time, sum of durations
n = len(rows); rows = rows[int(n * 0.4):]
t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
busy = 0; cs, ce = ev[0]
for s, e in ev[1:]:
    if s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
tot = sum(e - s for s, e in ev)
print('span %.2f ms  busy(union) %.2f ms  sum of kernel durations %.2f ms  -> idle %.1f%%  overlap factor %.2f' % ((t1 - t0) / 1e6, busy / 1e6, tot / 1e6, 100 * (1 - busy / (t1 - t0)), tot / busy))
PY
rm -rf gpurun_out/trace_$tag
tail -3 gpurun_out/timeline_$tag.txt

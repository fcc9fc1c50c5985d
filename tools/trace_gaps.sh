# usage: tools/trace_gaps.sh <tag> [bench args]
# rocprofv3 kernel trace of a short bench run (split pipeline as timed); the gaps between consecutive
# kernels of the wide-layer queue and a steady-state timeline window -> gpurun_out/gaps_<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $out/trace_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$tag -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 8 --no-cpu-baseline "$@" > $out/trace_$tag.log 2>&1 || { tail -5 $out/trace_$tag.log; exit 1; }
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/trace_$tag/*kernel_trace.csv gpurun_out/trace_$tag/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 tools/queue_trace.py $f > gpurun_out/gaps_$tag.txt
python3 - $f <<'PY' >> gpurun_out/gaps_$tag.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'(void )?(\w+)(<[^>]*>)?', n)
    return (m.group(2) + (m.group(3) or '')) if m else n[:40]
# a window of ~3 frames in the steady state (70 % into the trace)
i0 = int(len(rows) * 0.7)
t0 = int(rows[i0]['Start_Timestamp'])
print('--- timeline: start us, duration us, queue, kernel')
for r in rows[i0:]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    if s > 6500: break
    e = (int(r['End_Timestamp']) - t0) / 1e3
    print('%8.1f %7.1f  q%s %s' % (s, e - s, r.get('Queue_Id', '?'), short(r['Kernel_Name'])[:60]))
PY
gzip -c $f > gpurun_out/trace_$tag.csv.gz
rm -rf gpurun_out/trace_$tag
head -20 gpurun_out/gaps_$tag.txt
tail -1 gpurun_out/trace_$tag.log | cut -c1-300

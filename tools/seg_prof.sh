# usage: tools/seg_prof.sh <tag> [seg_bench args]   -- per-kernel durations with the GPU to each segment alone
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/segprof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/segprof_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/seg_bench.py "$@" > $out/segprof_$tag.log 2>&1 || { tail -5 $out/segprof_$tag.log; exit 1; }
cd $GRAFT_REPO_ROOT
cp $(ls gpurun_out/segprof_$tag/*kernel_stats.csv | head -1) gpurun_out/segprof_${tag}_kernel_stats.csv
rm -rf gpurun_out/segprof_$tag
grep alone gpurun_out/segprof_$tag.log
python3 - gpurun_out/segprof_${tag}_kernel_stats.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    m = re.match(r'(void )?(\w+)(<[^>]*>)?', n)
    return (m.group(2) + (m.group(3) or '')) if m else n[:40]
for r in rows:
    n = short(r['Name'])
    if n.startswith(('at', '__amd', 'spin')): continue
    print('%-52s calls %6s  avg %8.1f us  min %7.1f  max %7.1f' % (n[:52], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY

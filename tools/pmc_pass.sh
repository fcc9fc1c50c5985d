# usage: tools/pmc_pass.sh "<COUNTER ...>" <tag> [bench args]  -- one rocprofv3 counter pass over a short eager bench
# run, summed per kernel into gpurun_out/pmc_<tag>.csv (columns: kernel, dispatches, counters..., duration_ns)
ctr="$1"; tag="$2"; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmcraw_$tag
timeout -s KILL 150 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/pmcraw_$tag -- python3 bench.py --steps 3 --warmup 1 --frames 1 --no-graph --no-overlap --streams 1 --no-cpu-baseline "$@" > gpurun_out/pmc_$tag.log 2>&1 || { echo "pass failed"; tail -5 gpurun_out/pmc_$tag.log; exit 1; }
python3 - "$tag" <<'PY'
import csv, glob, sys, collections, re
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/pmcraw_{tag}/**/*counter_collection.csv', recursive=True)[0]
t = glob.glob(f'gpurun_out/pmcraw_{tag}/**/*kernel_trace.csv', recursive=True)[0]
dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(t))}
agg = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
names = []
for r in csv.DictReader(open(f)):
    k = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name']); k = re.sub(r'\(.*', '', k)
    c = r['Counter_Name']
    if c not in names: names.append(c)
    agg[k][c] += float(r['Counter_Value']); nd[k].add(r['Dispatch_Id'])
    agg[k]['_dur_' + r['Dispatch_Id']] = dur.get(r['Dispatch_Id'], 0)
with open(f'gpurun_out/pmc_{tag}.csv', 'w') as o:
    w = csv.writer(o); w.writerow(['kernel', 'dispatches'] + names + ['duration_ns'])
    for k in sorted(agg):
        d = sum(v for kk, v in agg[k].items() if kk.startswith('_dur_'))
        w.writerow([k, len(nd[k])] + [int(agg[k][c]) for c in names] + [int(d)])
print(open(f'gpurun_out/pmc_{tag}.csv').read())
PY
rm -rf gpurun_out/pmcraw_$tag

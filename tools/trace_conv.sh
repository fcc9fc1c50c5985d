# usage: tools/trace_conv.sh -- per-dispatch durations of the dominant conv kernel over a short bench run
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/trace_conv
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_conv -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_conv.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/trace_conv.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/trace_conv/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = [(int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r.get('Stream_Id', r.get('Queue_Id', '?')))
       for r in rows if 'wide_kernel<128, 128' in r['Kernel_Name']]
print('dispatches', len(sel))
t0 = sel[0][0]
for s, d, q in sel:
    print(f'{(s - t0) / 1e6:10.3f} ms  {d / 1e3:8.1f} us  q{q}')
PY
rm -rf gpurun_out/trace_conv

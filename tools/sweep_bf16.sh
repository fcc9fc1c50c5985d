# usage: tools/sweep_bf16.sh "tm,cg ..." "mb,d1 ..."   -- tile / direct variants of the bf16 conv
mkdir -p gpurun_out/sweep
run() {
  timeout -k 10 200 python bench.py --dtype bf16 --steps 30 --warmup 5 --no-cpu-baseline --detail gpurun_out/sweep/d.json > gpurun_out/sweep/b.log 2>&1 || { echo "== $1 FAILED"; tail -3 gpurun_out/sweep/b.log; return; }
  python - <<PY
import json
line=json.loads(open('gpurun_out/sweep/b.log').read().strip().splitlines()[-1])
d=json.load(open('gpurun_out/sweep/d.json'))
print('== %-22s %7.2f fps | ' % ('$1', line['value']) + '  '.join('%s %.3f'%(k,v['ms']) for k,v in d['groups'].items()), '| conv %.3f' % d['conv_ms_per_frame'])
PY
}
for cfg in $1; do SF_BF16_TILE=$cfg run "tile $cfg"; done
for cfg in $2; do SF_BF16_DIRECT=$cfg run "direct $cfg"; done

# usage: tools/sweep_direct.sh "mb,d1 ..." [test]
mkdir -p gpurun_out/sweep
for cfg in $1; do
  export SF_CONV_DIRECT=$cfg
  if [ -n "$2" ]; then
    timeout -k 10 200 python -u -m pytest tests/test_gpu_conv.py tests/test_golden_encoder.py -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -1
  fi
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --streams 1 --no-overlap --no-cpu-baseline --detail gpurun_out/sweep/d.json > gpurun_out/sweep/b.log 2>&1 || { echo "== $cfg FAILED"; tail -3 gpurun_out/sweep/b.log; continue; }
  python - <<PY
import json
line=json.loads(open('gpurun_out/sweep/b.log').read().strip().splitlines()[-1])
d=json.load(open('gpurun_out/sweep/d.json'))
print('== direct %-8s %7.2f fps | ' % ('$cfg', line['value']) + '  '.join('%s %.3f'%(k,v['ms']) for k,v in d['groups'].items()), '| conv %.3f' % d['conv_ms_per_frame'])
PY
done

# usage: tools/sweep_conv.sh "cfg1 cfg2 ..." "bal..." [test]   (cfg = tm,am,waves,abuf; abuf 3 = wide kernel)
mkdir -p gpurun_out/sweep
for cfg in $1; do
for bal in ${2:-0 1}; do
  export SF_CONV_CFG=$cfg SF_CONV_BALANCE=$bal
  if [ -n "$3" ]; then
    timeout -k 10 200 python -u -m pytest tests/test_gpu_conv.py tests/test_golden_encoder.py -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -1
  fi
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --detail gpurun_out/sweep/d.json > gpurun_out/sweep/b.log 2>&1 || { echo "== $cfg bal=$bal FAILED"; tail -3 gpurun_out/sweep/b.log; continue; }
  python - <<PY
import json
line=json.loads(open('gpurun_out/sweep/b.log').read().strip().splitlines()[-1])
d=json.load(open('gpurun_out/sweep/d.json'))
print('== %-12s bal=%s  %7.2f fps | ' % ('$cfg', '$bal', line['value']) + '  '.join('%s %.3f'%(k,v['ms']) for k,v in d['groups'].items() if int(k.split('x')[0])>=32), '| conv %.3f' % d['conv_ms_per_frame'])
PY
done
done

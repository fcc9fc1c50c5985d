mkdir -p gpurun_out/sweep
for cfg in 3,1,0 3,2,0; do
for bal in 0 1; do
  echo "== $cfg bal=$bal"
  export SF_CONV_CFG=$cfg SF_CONV_BALANCE=$bal
  timeout -k 10 200 python -u -m pytest tests/test_gpu_conv.py tests/test_golden_encoder.py -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -1
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --detail gpurun_out/sweep/d_$cfg.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  python - <<PY
import json
d=json.load(open('gpurun_out/sweep/d_$cfg.json'))
print('   '+'  '.join('%s %.3f'%(k,v['ms']) for k,v in d['groups'].items()))
PY
done
done

mkdir -p gpurun_out/sweep
for bal in 0 2 4 6 8 14; do
for cfg in 64,4,1; do
  echo "== $cfg flags=$bal"
  SF_CONV_BALANCE=$bal SF_CONV_CFG=$cfg timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --detail gpurun_out/sweep/d_$cfg.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  python - <<PY
import json
d=json.load(open('gpurun_out/sweep/d_$cfg.json'))
print('   '+'  '.join('%s %.3f'%(k,v['ms']) for k,v in d['groups'].items()))
PY
done
done

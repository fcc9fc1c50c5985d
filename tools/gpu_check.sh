# usage: tools/gpu_check.sh <tag>   -- conv/encoder parity tests, short bench with per-layer detail, stamp profile
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_gpu_conv.py tests/test_golden_encoder.py tests/test_gpu_pipeline.py -x -q --timeout 120 --timeout-method thread -m gpu 2>&1 | tail -3 &&
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --detail gpurun_out/d_$tag.json > gpurun_out/b_$tag.log 2>&1 &&
python - <<PY
import json
line=json.loads(open('gpurun_out/b_$tag.log').read().strip().splitlines()[-1])
print(line['value'], line['ms_per_step'], line['roofline']['achieved'])
d=json.load(open('gpurun_out/d_$tag.json'))
print('   '+'  '.join('%s %.3f'%(k,v['ms']) for k,v in d['groups'].items()), ' conv total %.3f' % d['conv_ms_per_frame'])
PY
timeout -k 10 200 python tools/prof_conv.py > gpurun_out/p_$tag.log 2>&1; tail -25 gpurun_out/p_$tag.log

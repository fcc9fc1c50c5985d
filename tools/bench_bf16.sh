# usage: tools/bench_bf16.sh <tag> [extra bench args]  -- bf16 bench with per-layer detail
tag=${1:-x}; shift
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --dtype bf16 --steps 40 --warmup 5 --no-cpu-baseline --detail gpurun_out/d16_$tag.json "$@" > gpurun_out/b16_$tag.log 2>&1 || { tail -5 gpurun_out/b16_$tag.log; exit 1; }
python - <<PY
import json
line=json.loads(open('gpurun_out/b16_$tag.log').read().strip().splitlines()[-1])
print(line['value'], line['ms_per_step'], line['roofline'])
d=json.load(open('gpurun_out/d16_$tag.json'))
print('   '+'  '.join('%s %.3f'%(k,v['ms']) for k,v in d['groups'].items()), ' conv total %.3f' % d['conv_ms_per_frame'])
for k,v in d['groups'].items():
    print('   %-10s %7.3f ms  %7.1f GB/s  %7.1f TF/s' % (k, v['ms'], v['bytes']/v['ms']/1e6, v['flops']/v['ms']/1e9))
PY

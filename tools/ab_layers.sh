# usage: tools/ab_layers.sh <tag> [bench.py options ...]   -- short bench with per-layer detail
tag=$1; shift
timeout -k 10 200 python bench.py "$@" --steps 60 --warmup 10 --no-cpu-baseline --detail gpurun_out/d_$tag.json > gpurun_out/b_$tag.log 2>&1 || { tail -5 gpurun_out/b_$tag.log; exit 1; }
python - <<PY
import json
line=json.loads(open('gpurun_out/b_$tag.log').read().strip().splitlines()[-1])
d=json.load(open('gpurun_out/d_$tag.json'))
print('$tag', line['value'], 'frames/s', line['ms_per_step'], 'ms |', '  '.join('%s %.1f'%(k,v['ms']/v['launches']*1e3) for k,v in d['groups'].items()), '| conv total %.3f ms' % d['conv_ms_per_frame'])
PY

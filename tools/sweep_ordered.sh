# usage: tools/sweep_ordered.sh -- end-to-end bench with the ordered kernel on different layer shapes
for sel in none 32x32 64x64 32x32,64x64 32x32,64x64,128x128 16x32,32x32,32x64,64x64 all; do
  echo "== --ordered $sel"
  timeout -k 10 200 python bench.py --ordered $sel --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | grep '^{"metric' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac'])" || exit 1
done

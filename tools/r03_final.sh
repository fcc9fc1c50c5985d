# usage: tools/r03_final.sh <round tag>   -- every profile / bench line of a round on one box, into gpurun_out/
tag=${1:-r03}
set -o pipefail
bash tools/r03_profiles.sh $tag f32 1 > gpurun_out/${tag}_final_a.log 2>&1 || { tail -5 gpurun_out/${tag}_final_a.log; exit 1; }
bash tools/r03_profiles.sh $tag bf16 1 > gpurun_out/${tag}_final_b.log 2>&1 || { tail -5 gpurun_out/${tag}_final_b.log; exit 1; }
bash tools/prof_bench.sh ${tag}_f32 > /dev/null || exit 1
bash tools/prof_bench.sh ${tag}_bf16 --dtype bf16 > /dev/null || exit 1
bash tools/prof_bench.sh ${tag}_bf16_b8 --dtype bf16 --batch 8 > /dev/null || exit 1
: > gpurun_out/${tag}_bench_lines.txt
timeout -k 10 400 python bench.py 2> /dev/null | tail -1 > gpurun_out/${tag}_bench_line.json || exit 1
cut -c1-160 gpurun_out/${tag}_bench_line.json
for args in "--dtype bf16" "--dtype bf16 --batch 8 --steps 50 --warmup 5" "--workload waymo" "--batch 8 --steps 30 --warmup 3" "--frame-pipeline"; do
  echo "== bench.py --no-cpu-baseline $args" >> gpurun_out/${tag}_bench_lines.txt
  timeout -k 10 300 python bench.py --no-cpu-baseline $args 2> /dev/null | tail -1 >> gpurun_out/${tag}_bench_lines.txt || exit 1
  tail -1 gpurun_out/${tag}_bench_lines.txt | cut -c1-160
done

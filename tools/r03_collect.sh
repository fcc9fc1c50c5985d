# usage: tools/r03_collect.sh <tag>  -- copy what tools/r03_final.sh left in gpurun_out/ into profiles/ (run here, after the GPU call)
tag=${1:-r03}
for f in stage_roofline.md stage_roofline.json stage_roofline_bf16.md stage_roofline_bf16.json pmc_hbm_traffic.json pmc_hbm_traffic_bf16.json segments_alone_kernel_stats.csv segments_alone_bf16_kernel_stats.csv bench_line.json bench_lines.txt; do
  cp gpurun_out/${tag}_$f profiles/ || exit 1
done
cp gpurun_out/prof_${tag}_f32_kernel_stats.csv profiles/${tag}_bench_nusc_f32_kernel_stats.csv
cp gpurun_out/prof_${tag}_bf16_kernel_stats.csv profiles/${tag}_bench_nusc_bf16_kernel_stats.csv
cp gpurun_out/prof_${tag}_bf16_b8_kernel_stats.csv profiles/${tag}_bench_nusc_bf16_b8_kernel_stats.csv
python tools/hbm_rates.py $tag > /dev/null

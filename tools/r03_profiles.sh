# usage: tools/r03_profiles.sh <round tag, e.g. r03> [dtype f32|bf16] [batch]
# One box, in order: (1) rocprofv3 kernel trace of the three segments alone (tools/seg_bench.py),
# (2) bench.py --detail (algorithmic work), (3) FETCH_SIZE / WRITE_SIZE passes over an eager run,
# (4) tools/stage_roofline.py -> gpurun_out/<tag>_stage_roofline[_<dtype>].md / .json
tag=$1; dt=${2:-f32}; b=${3:-1}
sfx=""; [ "$dt" != f32 ] && sfx="_$dt"; [ "$b" != 1 ] && sfx="${sfx}_b$b"
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/sr_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/sr_trace -o p -- python3 $GRAFT_REPO_ROOT/tools/seg_bench.py --iters 10 --dtype $dt --batch $b > $out/sr_trace.log 2>&1 || { tail -5 $out/sr_trace.log; exit 1; }
cd $GRAFT_REPO_ROOT
grep alone gpurun_out/sr_trace.log
timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --dtype $dt --batch $b --detail gpurun_out/${tag}_detail$sfx.json > gpurun_out/sr_bench.log 2>&1 || { tail -5 gpurun_out/sr_bench.log; exit 1; }
tail -1 gpurun_out/sr_bench.log | cut -c1-200
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/sr_$c
  timeout -s KILL 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/sr_$c -- python3 bench.py --steps 4 --warmup 0 --frames 4 --no-graph --no-overlap --streams 1 --no-cpu-baseline --dtype $dt --batch $b > gpurun_out/sr_$c.log 2>&1 || { echo "pass $c failed"; tail -5 gpurun_out/sr_$c.log; exit 1; }
done
python3 tools/stage_roofline.py --trace $(ls gpurun_out/sr_trace/*kernel_trace.csv | head -1) --detail gpurun_out/${tag}_detail$sfx.json \
  --fetch $(find gpurun_out/sr_FETCH_SIZE -name '*counter_collection.csv' | head -1) --write $(find gpurun_out/sr_WRITE_SIZE -name '*counter_collection.csv' | head -1) \
  --dtype $dt --title "nuScenes-shaped frame, $dt, batch $b" --out gpurun_out/${tag}_stage_roofline$sfx > gpurun_out/sr_table.log 2>&1 || { tail -20 gpurun_out/sr_table.log; exit 1; }
python3 tools/pmc_traffic.py gpurun_out/sr_FETCH_SIZE gpurun_out/sr_WRITE_SIZE > /dev/null && mv gpurun_out/pmc_traffic.json gpurun_out/${tag}_pmc_hbm_traffic$sfx.json
cp $(ls gpurun_out/sr_trace/*kernel_stats.csv | head -1) gpurun_out/${tag}_segments_alone${sfx}_kernel_stats.csv
rm -rf gpurun_out/sr_trace gpurun_out/sr_FETCH_SIZE gpurun_out/sr_WRITE_SIZE
cat gpurun_out/sr_table.log

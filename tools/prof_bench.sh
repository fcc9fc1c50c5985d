# usage: tools/prof_bench.sh <tag> [bench args]
# rocprofv3 --kernel-trace --stats of a bench.py run (graph replay, as timed) -> gpurun_out/prof_<tag>_kernel_stats.csv
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log; exit 1; }
cd $GRAFT_REPO_ROOT
cp $(ls gpurun_out/prof_$tag/*kernel_stats.csv | head -1) gpurun_out/prof_${tag}_kernel_stats.csv
tail -1 gpurun_out/prof_$tag.log | cut -c1-400
rm -rf gpurun_out/prof_$tag

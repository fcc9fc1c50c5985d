# usage: tools/prof_second.sh <tag> [bench_second args] -- rocprofv3 kernel stats of tools/bench_second.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_second.py --iters 10 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log; exit 1; }
cd $GRAFT_REPO_ROOT
cp $(ls gpurun_out/prof_$tag/*kernel_stats.csv | head -1) gpurun_out/prof_${tag}_kernel_stats.csv
tail -1 gpurun_out/prof_$tag.log
rm -rf gpurun_out/prof_$tag

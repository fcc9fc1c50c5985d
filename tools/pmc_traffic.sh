# HBM-side traffic of every kernel: two separate counter passes (FETCH_SIZE uses 3 TCC counters,
# WRITE_SIZE 2: they do not fit one pass), eager launches so that every dispatch is counted.
# usage: tools/pmc_traffic.sh [tag] [bench args]   -> gpurun_out/pmc_traffic<tag>.json
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 4 --warmup 0 --frames 4 --no-graph --no-overlap --streams 1 --no-cpu-baseline "$@" > gpurun_out/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -5 gpurun_out/pmc_$c.log; exit 1; }
done
python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE && mv gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic$tag.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE

# HBM-side traffic of every kernel: two separate counter passes (FETCH_SIZE uses 3 TCC counters,
# WRITE_SIZE 2: they do not fit one pass), eager launches so that every dispatch is counted.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 3 --warmup 1 --frames 1 --no-graph --no-overlap --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1 || { echo "pass $c failed"; tail -5 gpurun_out/pmc_$c.log; exit 1; }
done
python3 tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE

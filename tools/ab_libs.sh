# usage (on the GPU box): tools/ab_libs.sh <rounds> <bench args...>  -- bench.py with every library variant under ab/ in turn
# (variants are built out of tree by hand; the box's copy of the tree is scratch, the in-tree library is overwritten there only)
rounds=$1; shift
for r in $(seq $rounds); do
  for v in ab/lib_*.so; do
    cp $v sparsefusion_amd/libsparsefusion_hip.so
    val=$(timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])") || exit 1
    echo "$r $v $val"
  done
done

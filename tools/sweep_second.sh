# usage: tools/sweep_second.sh [batch]  -- rows-per-workgroup sweep of the dense conv2d kernel (SF_CONV2D_TH)
b=${1:-1}
for th in 0 2 3 4 6; do
  echo "== SF_CONV2D_TH=$th batch $b"
  SF_CONV2D_TH=$th timeout -k 10 120 python tools/bench_second.py --batch $b 2>&1 | grep -v amdgpu.ids || exit 1
done

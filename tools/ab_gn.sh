# Same-box A/B of the GroupNorm apply kernel variants (isolated bench_gn + the headline bench).
export PYTHONPATH=$PWD
OUT=${1:-gpurun_out/abg}; mkdir -p $OUT; shift
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "group_norm or groupnorm or gn" 2>&1 | tail -2
for lib in "$@"; do
  if [ $lib = default ]; then unset FLAIR_HIP_LIB; else export FLAIR_HIP_LIB=$PWD/$lib; fi
  echo "== $lib" | tee -a $OUT/gn.log
  timeout -k 10 200 python tools/bench_gn.py 2>&1 | head -3 | tee -a $OUT/gn.log
done
unset FLAIR_HIP_LIB
bash tools/ab_libs.sh $OUT 2 "$@"

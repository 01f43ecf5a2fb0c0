# Same-box A/B of the alignment kernel's epilogue (batched biases, 16-byte buffer stores) against the previous library.
export PYTHONPATH=$PWD
OUT=${1:-gpurun_out/abd}; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "dcn or deform or align" 2>&1 | tail -2
for lib in prev new prev new; do
  if [ $lib = prev ]; then export FLAIR_HIP_LIB=$PWD/tools/probes/libflair_prev.so; else unset FLAIR_HIP_LIB; fi
  echo "== $lib" | tee -a $OUT/dcn.log
  timeout -k 10 200 python tools/bench_dcn.py 2>&1 | tail -2 | tee -a $OUT/dcn.log
done
unset FLAIR_HIP_LIB
bash tools/ab_libs.sh $OUT 3 tools/probes/libflair_prev.so default

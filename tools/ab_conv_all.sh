# Same-box A/B of every conv shape of one forward (tools/bench_conv.py) between a previous library build and the in-tree one.
export PYTHONPATH=$PWD
OUT=${1:-gpurun_out/abe2}; mkdir -p $OUT
for lib in prev new prev new; do
  if [ $lib = prev ]; then export FLAIR_HIP_LIB=$PWD/tools/probes/libflair_prev.so; else unset FLAIR_HIP_LIB; fi
  echo "== $lib" >> $OUT/conv.log
  timeout -k 10 200 python tools/bench_conv.py bf16 frame 2>&1 >> $OUT/conv.log
done
cat $OUT/conv.log

# Same-box A/B of the batched-residual epilogues (LDS-DMA conv kernel, K-split per-frame kernel) against the previous
# library (tools/probes/libflair_prev.so = HEAD~ build): kernel tests, isolated shapes with / without residual, the bench.
set -o pipefail
export PYTHONPATH=$PWD
OUT=${1:-gpurun_out/abe}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q -m gpu > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for lib in prev new prev new; do
  for res in 0 1; do
    for shape in "L1 frame 128->128" "L0 frame 64->64" "L1 frame 384->128" "L0 clip 64->64 2d"; do
      unset FLAIR_HIP_LIB
      if [ $lib = prev ]; then export FLAIR_HIP_LIB=$PWD/tools/probes/libflair_prev.so; fi
      echo -n "$lib res=$res  " >> $OUT/conv.log
      FLAIR_BENCH_RES=$res timeout -k 10 120 python tools/bench_conv.py bf16 "$shape" 2>&1 | tail -1 >> $OUT/conv.log
    done
  done
done
cat $OUT/conv.log
unset FLAIR_HIP_LIB
for lib in prev new prev new prev new; do
  if [ $lib = prev ]; then export FLAIR_HIP_LIB=$PWD/tools/probes/libflair_prev.so; else unset FLAIR_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>>$OUT/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(l['ms_per_step'],2), 'ms/step')" | tee -a $OUT/bench.log
done

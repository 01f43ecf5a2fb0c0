export PYTHONPATH=$PWD
export FLAIR_HIP_LIB=$PWD/tools/probes/libflair_timing.so
O=${1:-gpurun_out/r3g}; mkdir -p $O
for shape in "L0 clip 64->64 2d" "L1 clip 128->128 3d"; do
  for mode in 0 13 15 16; do
    echo -n "dma  mode $mode  " >> $O/switches.txt
    FLAIR_CONV_DEBUG=$mode timeout -k 5 60 python tools/bench_conv.py bf16 "$shape" 2>/dev/null | grep -v amdgpu >> $O/switches.txt
  done
done
cat $O/switches.txt

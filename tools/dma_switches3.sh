export PYTHONPATH=$PWD
export FLAIR_HIP_LIB=$PWD/tools/probes/libflair_timing.so
O=${1:-gpurun_out/r3l}; mkdir -p $O
for shape in "L0 frame 64->64" "L0 frame 224->64" "L1 frame 128->128" "L1 frame 416->128"; do
  for mode in 0 11 12 13; do
    echo -n "dma-frame mode $mode  " >> $O/switches.txt
    FLAIR_CONV_DEBUG=$mode timeout -k 5 60 python tools/bench_conv.py bf16 "$shape" 2>/dev/null | grep -v amdgpu >> $O/switches.txt
  done
done
cat $O/switches.txt

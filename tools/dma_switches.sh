# Phase-timing switches of the persistent LDS-DMA conv kernel (diagnostic build: make -C flair_amd/csrc timing):
# 0 full kernel, 11 no MFMA phase, 12 no DMA (first chunk only), 13 no epilogue.  Usage: bash tools/dma_switches.sh <outdir>
export PYTHONPATH=$PWD
export FLAIR_HIP_LIB=$PWD/tools/probes/libflair_timing.so
O=${1:-gpurun_out/dma_sw}
mkdir -p $O
for shape in "L0 clip 64->64 2d" "L0 clip 64->64 3d" "L1 clip 128->128 3d" "L1 frame 128->432" "L0 frame 64->432"; do
  for mode in 0 11 12 13; do
    echo -n "mode $mode  " >> $O/switches.txt
    FLAIR_CONV_DEBUG=$mode timeout -k 5 60 python tools/bench_conv.py bf16 "$shape" 2>/dev/null | grep -v amdgpu >> $O/switches.txt
  done
done
cat $O/switches.txt

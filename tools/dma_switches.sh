# Phase-timing switches of the conv kernels (diagnostic build: make -C flair_amd/csrc timing).
#   persistent LDS-DMA kernel: 0 full, 11 no MFMA phase, 12 no DMA (first chunk only), 13 no epilogue, 14 epilogue stores waited for
#   halo kernel (FLAIR_CONV_DMA=0): 0 full, 1 no MFMA phase, 2 no in-loop reloads, 5 no epilogue
# Usage: bash tools/dma_switches.sh <outdir>
export PYTHONPATH=$PWD
export FLAIR_HIP_LIB=$PWD/tools/probes/libflair_timing.so
O=${1:-gpurun_out/dma_sw}
mkdir -p $O
for shape in "L0 clip 64->64 2d" "L0 clip 64->64 3d" "L1 clip 128->128 3d" "L1 frame 128->432" "L0 frame 64->432"; do
  for mode in 0 11 12 13 14; do
    echo -n "dma  mode $mode  " >> $O/switches.txt
    FLAIR_CONV_DEBUG=$mode timeout -k 5 60 python tools/bench_conv.py bf16 "$shape" 2>/dev/null | grep -v amdgpu >> $O/switches.txt
  done
  for mode in 0 1 2 5; do
    echo -n "halo mode $mode  " >> $O/switches.txt
    FLAIR_CONV_DMA=0 FLAIR_CONV_DEBUG=$mode timeout -k 5 60 python tools/bench_conv.py bf16 "$shape" 2>/dev/null | grep -v amdgpu >> $O/switches.txt
  done
done
cat $O/switches.txt

# Phase-timing switches of the per-frame K-split conv kernel (diagnostic build): 0 full, 21 no MFMA phase, 22 no epilogue,
# 23 no in-loop staging, 24 empty kernel (launch + setup only).  Usage: bash tools/ks_switches.sh <outdir>
export PYTHONPATH=$PWD
export FLAIR_HIP_LIB=$PWD/tools/probes/libflair_timing.so
O=${1:-gpurun_out/ks_sw}
mkdir -p $O
for shape in "L0 frame 64->64" "L0 frame 224->64" "L1 frame 128->128" "L1 frame 416->128"; do
  for mode in 0 21 22 23 24; do
    echo -n "ks mode $mode  " >> $O/ks_switches.txt
    FLAIR_CONV_DEBUG=$mode timeout -k 5 60 python tools/bench_conv.py bf16 "$shape" 2>/dev/null | grep -v amdgpu >> $O/ks_switches.txt
  done
done
cat $O/ks_switches.txt

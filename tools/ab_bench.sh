# Same-box A/B of the headline bench: default vs the round-4 kernels off (= round 3) vs the round-3 LDS-DMA kernels off too vs every switch off
# (= the round-1 kernels and launch structure).  Boxes differ by up to 10 %: only same-call numbers compare.
set -o pipefail
export PYTHONPATH=$PWD
OUT=${1:-gpurun_out/ab}
mkdir -p $OUT
run() {
  env "$@" timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>>$OUT/err.log | python -c "
import json,sys
l=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '->', round(l['ms_per_step'],2), 'ms/step', round(l['value'],4), 'frames/s')" | tee -a $OUT/ab.log
}
run FLAIR_NOOP=1
run FLAIR_CONV_RESIDENT=0 FLAIR_DCN_DOT2=0 FLAIR_DCN_TILE_C128=32 FLAIR_DMA_TFAST=0 FLAIR_DMA_RES_PREFETCH=1 FLAIR_DCN_PFD1=0
run FLAIR_CONV_DMA=0 FLAIR_CONV_DMA_FRAME=0 FLAIR_CONV_RESIDENT=0 FLAIR_DCN_DOT2=0 FLAIR_DCN_TILE_C128=32 FLAIR_DMA_TFAST=0 FLAIR_DMA_RES_PREFETCH=1 FLAIR_DCN_PFD1=0
run FLAIR_CHAIN=0 FLAIR_DCN_ACT=0 FLAIR_GN_BLOCKS=4096 FLAIR_FLOW2_CACHE=0 FLAIR_GN_FUSED_MAX=0 FLAIR_ATTN_V2=0 FLAIR_DEEPK_TILE128=0 FLAIR_CONV_LDS_SWZ=0 FLAIR_CONV_DMA=0 FLAIR_CONV_DMA_FRAME=0 FLAIR_CONV_RESIDENT=0 FLAIR_DCN_DOT2=0 FLAIR_DCN_TILE_C128=32 FLAIR_DMA_TFAST=0 FLAIR_DMA_RES_PREFETCH=1 FLAIR_DCN_PFD1=0
run FLAIR_NOOP=1
run FLAIR_CHAIN=0 FLAIR_DCN_ACT=0 FLAIR_GN_BLOCKS=4096 FLAIR_FLOW2_CACHE=0 FLAIR_GN_FUSED_MAX=0 FLAIR_ATTN_V2=0 FLAIR_DEEPK_TILE128=0 FLAIR_CONV_LDS_SWZ=0 FLAIR_CONV_DMA=0 FLAIR_CONV_DMA_FRAME=0 FLAIR_CONV_RESIDENT=0 FLAIR_DCN_DOT2=0 FLAIR_DCN_TILE_C128=32 FLAIR_DMA_TFAST=0 FLAIR_DMA_RES_PREFETCH=1 FLAIR_DCN_PFD1=0

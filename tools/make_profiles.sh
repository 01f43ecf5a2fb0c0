#!/bin/bash
# Regenerates the judged artifacts of one checkpoint on the GPU box:  bash tools/make_profiles.sh r01e
# (bench line, rocprofv3 kernel stats of the 6-step bench, PMC traffic passes, other task benches)
set -o pipefail
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$PWD}
export PYTHONPATH=$R
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout -k 10 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err
tail -1 $OUT/bench_full.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof /tmp/pmc_fetch /tmp/pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o p --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_profiled_steps6.json 2>/dev/null
cp /tmp/prof/p_kernel_stats.csv $OUT/bench_steps6_kernel_stats.csv
# HBM traffic of the dominant kernel: its shapes launched as often as one step launches them and nothing else (the whole bench under
# counter collection serialises ~10 000 launches and did not finish within the box's silence limit on two of three boxes)
timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_fetch -o b --output-format csv -- python $R/tools/pmc_dominant.py $OUT/bench_full.json > /dev/null 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_write -o b --output-format csv -- python $R/tools/pmc_dominant.py $OUT/bench_full.json > /dev/null 2>&1
python $R/tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $OUT/hbm_traffic_pmc.json > $OUT/hbm_traffic_pmc.txt
cd $R
# round 4: HBM traffic of EVERY kernel of the steady-state steps (FLAIR_BENCH_REPLAY=off: nothing but the steps' own launches; 14 s per pass)
bash tools/pmc_whole_step.sh gpurun_out/$TAG/pmc_whole 300 > /dev/null 2>&1
cp $OUT/pmc_whole/hbm_traffic_pmc_whole_step.txt $OUT/pmc_whole/hbm_traffic_pmc_whole_step.json $OUT/pmc_whole/passes.log $OUT/ 2>/dev/null
# the literal 16,32,64 attention layout of SURVEY.md section 3.2 beside the benched one
timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --attention-resolutions 16,32,64 > $OUT/bench_attn_16_32_64.json 2>/dev/null
tail -1 $OUT/bench_attn_16_32_64.json | cut -c1-200
python tools/bench_conv6.py > $OUT/conv6_isolated.txt 2>/dev/null
bash tools/pmc_dcn.sh gpurun_out/$TAG/dcn_pmc_sq.txt > /dev/null 2>&1
timeout -k 10 300 python bench.py --task x8_bicubic --no-cpu-baseline > $OUT/bench_x8_bicubic.json 2>/dev/null
timeout -k 10 300 python bench.py --task jpeg --no-cpu-baseline > $OUT/bench_jpeg.json 2>/dev/null
tail -1 $OUT/bench_x8_bicubic.json | cut -c1-200
tail -1 $OUT/bench_jpeg.json | cut -c1-200
python tools/bench_attn.py --json $OUT/attn_isolated.json > $OUT/attn_isolated.txt 2>/dev/null
python tools/bench_gn.py > $OUT/gn_isolated.txt 2>/dev/null
python tools/bench_dcn.py > $OUT/dcn_isolated.txt 2>/dev/null
python tools/bench_chain.py > $OUT/chain_isolated.txt 2>/dev/null
bash tools/ab_bench.sh $OUT/ab > /dev/null 2>&1
cat $OUT/ab/ab.log
# rows 8f-1 / 8f-4: the CodeFormer prior alone, and the 512^2 x 10-frame step without / with it
timeout -k 10 300 python tools/bench_codeformer.py --json $OUT/codeformer.json > $OUT/codeformer.txt 2>/dev/null
rm -f $OUT/bench_512x10_aux.jsonl
for aux in "--no-cpu-baseline --aux identity" "--aux codeformer" "--no-cpu-baseline --aux codeformer --aux-dtype bf16"; do   # the f32 line carries the CPU legs
  timeout -k 10 900 python bench.py --size 512 --frames 10 --steps 6 --warmup 1 $aux 2>/dev/null | tail -1 >> $OUT/bench_512x10_aux.jsonl
done
cut -c1-220 $OUT/bench_512x10_aux.jsonl
bash tools/run_trace_all.sh gpurun_out/$TAG/trace > /dev/null 2>&1
head -30 $OUT/trace/by_grid.txt

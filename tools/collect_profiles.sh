#!/bin/bash
# Collects the round's profile artifacts on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/...
# One rocprofv3 pass per counter group (PMC passes carry --kernel-trace only), the program itself after `--`.
set -e
TAG=${1:-x}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-inference --no-configs --no-feed --no-p16"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-inference --no-configs --no-feed --no-p16 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o f -- $B > /dev/null 2> $OUT/f.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o w -- $B > /dev/null 2> $OUT/w.err
echo "write done"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/c -o c -- $B > /dev/null 2> $OUT/c.err
echo "clock done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/s1 -o s -- $B > /dev/null 2> $OUT/s1.err
echo "sq1 done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/s2 -o s -- $B > /dev/null 2> $OUT/s2.err
echo "sq2 done"
python tools/pmc_traffic.py $OUT/f f $OUT/w w $OUT/pmc_traffic_all.json > $OUT/pmc_traffic.txt
python tools/pmc_clock.py $OUT/c c > $OUT/pmc_clock_mfma_util.txt
python tools/pmc_sq_ratios.py $OUT/s1 s $OUT/s2 s > $OUT/pmc_sq_ratios.txt
ls $OUT

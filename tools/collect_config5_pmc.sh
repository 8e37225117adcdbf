#!/bin/bash
# PMC passes over the config-5 run (MobileNetV3 forward, bs 256): gpurun_out/prof_c5/...   bash tools/collect_config5_pmc.sh
set -e
OUT=gpurun_out/prof_c5
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 tools/run_config5.py --batch 256 --reps 2 --cpu-batch 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o b -- $B > $OUT/run.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o f -- $B > /dev/null 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o w -- $B > /dev/null 2> $OUT/w.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/s1 -o s -- $B > /dev/null 2> $OUT/s1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/s2 -o s -- $B > /dev/null 2> $OUT/s2.err
python3 tools/pmc_traffic.py $OUT/f f $OUT/w w $OUT/pmc_traffic_all.json > $OUT/pmc_traffic.txt
python3 tools/pmc_sq_ratios.py $OUT/s1 s $OUT/s2 s > $OUT/pmc_sq_ratios.txt
ls $OUT

#!/bin/bash
# usage (ON THE GPU BOX, from the repo root): tools/prof_fp.sh <tag>   -- SQ counters of the 1024-frame step for whatever dispatch the environment selects
TAG=${1:?tag}
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $B --steps 5 --warmup 2 > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -o q -- python3 $B --steps 2 --warmup 1 > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -o q -- python3 $B --steps 2 --warmup 1 > $OUT/sq2.log 2>&1
cd $R
python tools/pmc_summary.py $(find $OUT/sq1 -name "*counter_collection.csv" | head -1) $(find $OUT/sq2 -name "*counter_collection.csv" | head -1) > $OUT/${TAG}_sq.txt
grep -E "k_fp|k_pre" $(find $OUT/stats -name "*kernel_stats.csv" | head -1) | cut -c1-200 > $OUT/${TAG}_stats.txt
cat $OUT/${TAG}_stats.txt; grep -A17 "k_fp_" $OUT/${TAG}_sq.txt

#!/bin/bash
# Collects everything profiles/ holds for one code state.  Run ON THE GPU BOX from the repo root:
#     tools/collect_profiles.sh <tag>        -> gpurun_out/<tag>/{stats,fetch,write,sq1,sq2}/..., gpurun_out/<tag>/bench.json
# Each rocprofv3 call is its own process with the program directly after `--`; counters are collected in their own
# passes with nothing but --pmc (never together with a trace option), FETCH_SIZE and WRITE_SIZE separately.
set -e
TAG=${1:?tag}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -o q -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -o q -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/sq2.log 2>&1
cd $R
find $OUT -name "*.csv" | head -20
# the label-masked variant, the producers and the consumer of the path (kernel stats only)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lc -o lc -- python3 $R/tools/time_labeled.py > $OUT/lc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n1 -o n1 -- python3 $R/tools/time_norm.py > $OUT/n1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n2 -o n2 -- python3 $R/tools/time_project.py > $OUT/n2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n3 -o n3 -- python3 $R/tools/time_slic.py > $OUT/n3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n4 -o n4 -- python3 $R/tools/time_stereo.py > $OUT/n4.log 2>&1
cd $R
python bench.py --batch1 --no-cpu-baseline > $OUT/bench_batch1.json 2>> $OUT/bench.err
tail -n 3 $OUT/lc.log $OUT/n1.log $OUT/n2.log $OUT/n3.log $OUT/n4.log

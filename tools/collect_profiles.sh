#!/bin/bash
# Collects everything profiles/ holds for one code state.  Run ON THE GPU BOX from the repo root:
#     tools/collect_profiles.sh <tag>        -> gpurun_out/<tag>/...  and the summaries in gpurun_out/<tag>/summary/
# Each rocprofv3 call is its own process with the program directly after `--`; counters are collected in their own
# passes with nothing but --pmc (never together with a trace option), FETCH_SIZE and WRITE_SIZE separately.
set -e
TAG=${1:?tag}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
SUM=$OUT/summary
mkdir -p $OUT $SUM
python bench.py > $SUM/${TAG}_bench.json 2> $OUT/bench.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $B --steps 5 --warmup 2 > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $B --steps 2 --warmup 1 > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $B --steps 2 --warmup 1 > $OUT/write.log 2>&1
echo "traffic done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq1 -o q -- python3 $B --steps 2 --warmup 1 > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq2 -o q -- python3 $B --steps 2 --warmup 1 > $OUT/sq2.log 2>&1
echo "sq done"
# the label-masked variant (BASELINE configs 2 / 3): kernel stats of both, then per config the two traffic passes and the first SQ pass
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lc -o lc -- python3 $R/tools/time_labeled.py > $OUT/lc.log 2>&1
python3 $R/tools/time_labeled.py > $OUT/lc_plain.log 2>&1      # the throughput line without the profiler attached
for c in 2 3; do
export LC_CONFIG=$c
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/lc${c}_fetch -o f -- python3 $R/tools/time_labeled.py > $OUT/lc${c}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/lc${c}_write -o w -- python3 $R/tools/time_labeled.py > $OUT/lc${c}_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/lc${c}_sq1 -o q -- python3 $R/tools/time_labeled.py > $OUT/lc${c}_sq1.log 2>&1
done
unset LC_CONFIG
echo "lc done"
# the producers and the consumer of the path (kernel stats only)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n1 -o n1 -- python3 $R/tools/time_norm.py > $OUT/n1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n2 -o n2 -- python3 $R/tools/time_project.py > $OUT/n2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n3 -o n3 -- python3 $R/tools/time_slic.py > $OUT/n3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/n4 -o n4 -- python3 $R/tools/time_stereo.py > $OUT/n4.log 2>&1
echo "n-rows done"
cd $R
python bench.py --batch1 --no-cpu-baseline --no-configs > $SUM/${TAG}_bench_batch1.json 2>> $OUT/bench.err
# ---- summaries (what gets copied into profiles/)
f() { find $OUT/$1 -name "*$2" | head -1; }
cp "$(f stats kernel_stats.csv)" $SUM/${TAG}_kernel_stats.csv
cp "$(f lc kernel_stats.csv)" $SUM/${TAG}_lc_kernel_stats.csv
for n in n1 n2 n3 n4; do cp "$(f $n kernel_stats.csv)" $SUM/${TAG}_${n}_kernel_stats.csv; done
# the N-row throughput lines without the profiler attached
python tools/time_norm.py > $SUM/${TAG}_n1_tool_output.txt 2>/dev/null; python tools/time_project.py > $SUM/${TAG}_n2_tool_output.txt 2>/dev/null
python tools/time_slic.py > $SUM/${TAG}_n3_tool_output.txt 2>/dev/null; python tools/time_stereo.py > $SUM/${TAG}_n4_tool_output.txt 2>/dev/null
grep -E "frames/s" $OUT/lc_plain.log > $SUM/${TAG}_lc_tool_output.txt || true
python tools/pmc_traffic.py "$(f fetch counter_collection.csv)" "$(f write counter_collection.csv)" 8 $SUM/${TAG}_traffic.json > /dev/null
for c in 2 3; do
python tools/pmc_traffic.py "$(f lc${c}_fetch counter_collection.csv)" "$(f lc${c}_write counter_collection.csv)" 7 $SUM/${TAG}_lc_config${c}_traffic.json > /dev/null
python tools/pmc_summary.py "$(f lc${c}_sq1 counter_collection.csv)" > $SUM/${TAG}_lc_config${c}_sq_counters.txt
done
python tools/pmc_valu.py "$(f sq1 counter_collection.csv)" $SUM/${TAG}_valu.json > /dev/null
python tools/pmc_summary.py "$(f sq1 counter_collection.csv)" "$(f sq2 counter_collection.csv)" > $SUM/${TAG}_sq_counters.txt
ls -la $SUM
tail -n 3 $OUT/lc.log

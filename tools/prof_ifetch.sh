#!/bin/bash
# usage (ON THE GPU BOX, from the repo root): tools/prof_ifetch.sh <tag>  -- instruction-fetch / instruction-cache counters of the 1024-frame step
TAG=${1:?tag}
R=$(pwd); OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i -o "SQ[C]*_[A-Z_0-9]*\(IFETCH\|ICACHE\|INST_CACHE\)[A-Z_0-9]*" $OUT/counters.txt | sort -u > $OUT/ifetch_names.txt
cat $OUT/ifetch_names.txt
B="$R/bench.py --no-cpu-baseline --no-configs"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/i1 -o q -- python3 $B --steps 2 --warmup 1 > $OUT/i1.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/i2 -o q -- python3 $B --steps 2 --warmup 1 > $OUT/i2.log 2>&1
cd $R
python - <<P
import csv, glob, collections
for d in ('i1','i2'):
    fs = glob.glob('$OUT/'+d+'/**/*counter_collection.csv', recursive=True)
    if not fs: print(d, 'no file'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in acc.items():
        if 'k_fp_q' in k or 'k_pre_p' in k: print(k, {a: round(b / 1e6, 2) for a, b in v.items()})
P
tail -3 $OUT/i1.log $OUT/i2.log

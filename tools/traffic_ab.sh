#!/bin/bash
# usage (GPU box): tools/traffic_ab.sh abv/a.so abv/b.so ...  -- FETCH_SIZE / WRITE_SIZE passes of the 1024-frame step for every library variant
LIB=depth_completion_mt_amd/csrc/libdcmt_hip.so
R=$(pwd); cp $LIB /tmp/libdcmt_hip.keep
for v in "$@"; do
    cp "$v" $LIB; n=$(basename $v .so); OUT=$R/gpurun_out/traffic_$n; mkdir -p $OUT
    (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-configs --steps 2 --warmup 1 > $OUT/f.log 2>&1)
    (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/bench.py --no-cpu-baseline --no-configs --steps 2 --warmup 1 > $OUT/w.log 2>&1)
    python tools/pmc_traffic.py "$(find $OUT/fetch -name '*counter_collection.csv' | head -1)" "$(find $OUT/write -name '*counter_collection.csv' | head -1)" 8 $OUT/traffic.json > /dev/null
    python -c "
import json; t=json.load(open('$OUT/traffic.json'))
print('$n', round(t['hbm_bytes_per_step']/1e9,3), {k.split('::')[1][:28]:(round(x['read_bytes_per_step']/1e9,3),round(x['write_bytes_per_step']/1e9,3)) for k,x in t['kernels'].items() if x['read_bytes_per_step']>1e6})"
done
cp /tmp/libdcmt_hip.keep $LIB

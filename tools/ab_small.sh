#!/bin/bash
# usage (GPU box): tools/ab_small.sh FRAMES abv/a.so abv/b.so ...  -- bench.py at FRAMES frames per step for every library variant, twice, alternating
LIB=depth_completion_mt_amd/csrc/libdcmt_hip.so
N=$1; shift
cp $LIB /tmp/libdcmt_hip.keep
for rep in 1 2; do
for v in "$@"; do
    cp "$v" $LIB; n=$(basename $v .so)
    python bench.py --total-frames $N --no-configs --no-cpu-baseline > gpurun_out/ab_${n}_${N}_$rep.json 2> gpurun_out/ab_${n}_${N}_$rep.err || { cp /tmp/libdcmt_hip.keep $LIB; exit 1; }
    python - <<P
import json
d=json.load(open('gpurun_out/ab_${n}_${N}_$rep.json')); print('$n', $N, round(d['value']), round(d['ms_per_step'],4), {k:(round(v['ms'],4) if isinstance(v,dict) else round(v,4)) for k,v in d['roofline']['per_kernel'].items()}, flush=True)
P
done
done
cp /tmp/libdcmt_hip.keep $LIB

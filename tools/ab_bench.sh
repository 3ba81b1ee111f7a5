#!/bin/bash
# usage: tools/ab_bench.sh "<bench.py flags>" variants/a.so variants/b.so ...   -- whole-step A/B on ONE box: every prebuilt
# library variant through bench.py (frames/s and ms per step), three rounds; the in-tree library is restored at the end
FLAGS=$1; shift
LIB=depth_completion_mt_amd/csrc/libdcmt_hip.so
cp $LIB /tmp/libdcmt_hip.keep
for rep in 1 2 3; do
for v in "$@"; do
    cp "$v" $LIB
    echo "$v [$FLAGS]: $(python bench.py --no-configs --no-cpu-baseline $FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])")"
done
done
cp /tmp/libdcmt_hip.keep $LIB

#!/bin/bash
# usage: tools/sweep_variants.sh variants/*.so   -- times every prebuilt library variant with tools/time_stages.py
# (run on the GPU box; the variant files travel with the snapshot, the in-tree library is restored at the end)
set -e
LIB=depth_completion_mt_amd/csrc/libdcmt_hip.so
cp $LIB /tmp/libdcmt_hip.keep
for v in "$@"; do
    cp "$v" $LIB
    for i in 1 2; do echo "$v: $(python tools/time_stages.py 2>/dev/null | tail -1)"; done
done
cp /tmp/libdcmt_hip.keep $LIB

#!/bin/bash
# usage: [B=...] tools/sweep_variants.sh variants/*.so   -- times every prebuilt library variant (tools/time_variants.py: the library's own
# per-kernel events); run on the GPU box: the variant files travel with the snapshot, the in-tree library is restored at the end
LIB=depth_completion_mt_amd/csrc/libdcmt_hip.so
cp $LIB /tmp/libdcmt_hip.keep
for rep in 1 2; do
for v in "$@"; do
    cp "$v" $LIB
    echo "$v: $(python tools/time_variants.py 2>/dev/null | tail -1)"
done
done
cp /tmp/libdcmt_hip.keep $LIB

#!/bin/bash
# usage: [ENV=...] tools/sweep_variants.sh variants/*.so   -- times every prebuilt library variant with bench.py (frames/s, ms/step,
# live per-kernel split); run on the GPU box: the variant files travel with the snapshot, the in-tree library is restored at the end
set -e
LIB=depth_completion_mt_amd/csrc/libdcmt_hip.so
cp $LIB /tmp/libdcmt_hip.keep
for rep in 1 2 3; do
for v in "$@"; do
    cp "$v" $LIB
    echo "$v: $(python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],3), {k: round(v,3) for k,v in d["per_kernel_ms"].items()})')"
done
done
cp /tmp/libdcmt_hip.keep $LIB

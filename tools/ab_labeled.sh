#!/bin/bash
# usage (GPU box): tools/ab_labeled.sh abv/a.so abv/b.so ...  -- tools/time_labeled.py for every library variant, three times, alternating
LIB=depth_completion_mt_amd/csrc/libdcmt_hip.so
cp $LIB /tmp/libdcmt_hip.keep
for rep in 1 2 3; do for v in "$@"; do cp "$v" $LIB; echo "== $(basename $v .so)"; python tools/time_labeled.py 2>&1 | grep labeled | cut -c1-200; done; done
cp /tmp/libdcmt_hip.keep $LIB

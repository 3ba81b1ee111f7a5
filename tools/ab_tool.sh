#!/bin/bash
# usage: tools/ab_tool.sh tools/<script>.py variants/a.so variants/b.so ...   -- runs one timing script with every prebuilt
# library variant on ONE box (two rounds); the in-tree library is restored at the end
T=$1; shift
LIB=depth_completion_mt_amd/csrc/libdcmt_hip.so
cp $LIB /tmp/libdcmt_hip.keep
for rep in 1 2; do
for v in "$@"; do
    cp "$v" $LIB
    python $T 2>/dev/null | sed "s|^|$v: |"
done
done
cp /tmp/libdcmt_hip.keep $LIB

#!/bin/bash
# Builds tests/mock_opencv/shim_test against include/img_completion.h, the cv::Mat stand-in and
# libdcmt_hip.so.  --compile-only: just check that the shim compiles (no GPU needed).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
LIB="$ROOT/depth_completion_mt_amd/csrc"
g++ -std=c++11 -O1 -Wall -I"$ROOT/include" -I"$HERE" -c "$HERE/shim_test.cpp" -o "$HERE/shim_test.o"
if [ "$1" != "--compile-only" ]; then
  g++ "$HERE/shim_test.o" -o "$HERE/shim_test" -L"$LIB" -ldcmt_hip -Wl,-rpath,"$LIB" -Wl,-rpath,/opt/rocm/lib
fi

#!/bin/bash
# Round-3 session 16: wino4_f32 - cache policy of the patch loads
cd "$(dirname "$0")/../.."
for v in w4a1 w4a2 w4a3 w4a16 w4a17; do
  echo "=== $v"
  for shp in "32 92 92 256 256 0 1" "32 46 46 512 512 0 1"; do
    RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$v.so timeout 300 python tools/bench_conv3.py $shp 2>&1 | grep -v amdgpu.ids
  done
done

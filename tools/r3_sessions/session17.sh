#!/bin/bash
# Round-3 session 17: wino4_f32 - where in the chunk the patch loads are issued
cd "$(dirname "$0")/../.."
for v in w4l2 w4l12; do
  echo "=== $v"
  for shp in "32 92 92 256 256 0 1" "32 46 46 512 512 0 1"; do
    RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$v.so timeout 300 python tools/bench_conv3.py $shp 2>&1 | grep -v amdgpu.ids
  done
done

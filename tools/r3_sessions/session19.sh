#!/bin/bash
# Round-3 session 19: wino4_f32 (frequency-split waves) - ablations (results of the ablation builds are wrong, timing only)
cd "$(dirname "$0")/../.."
for v in base w4ns w4nb w4n3 w4nl w4l2 w4l4; do
  echo "=== $v"
  for shp in "32 92 92 256 256 0 1" "32 46 46 512 512 0 1"; do
    RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$v.so timeout 300 python tools/bench_conv3.py $shp 2>&1 | grep -v amdgpu.ids | sed -E 's/direct.*(f4 [0-9.]+ ms).*/\1/'
  done
done

#!/bin/bash
# Round-3 session 14: what each stream of wino4_f32 costs (ablation builds: results are wrong, timing only)
cd "$(dirname "$0")/../.."
for v in base w4ns w4nb w4na w4n3; do
  echo "=== $v"
  for shp in "32 92 92 256 256 0 1" "32 46 46 512 512 0 1" "32 184 184 128 128 1 1"; do
    RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$v.so timeout 300 python tools/bench_conv3.py $shp 2>&1 | grep -v amdgpu.ids
  done
done

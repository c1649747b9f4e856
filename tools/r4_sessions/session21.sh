#!/bin/bash
# Round-4 session 21: F(4x4,3x3) kernel - timing-only: patch loads as they would be on [channels / 8][pixels][8] planes
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for v in dev w4planes dev w4planes; do
  echo "=== $v"
  RTPOSE_LIB_PATH=$R/tools/exp/lib_$v.so python tools/profile_layers.py 32 368 368 3 fp32 2>&1 | grep -E "^model0|^model1_1.[024]|sum of|^k="
done > $O/s21_w4planes.txt 2>&1
cat $O/s21_w4planes.txt

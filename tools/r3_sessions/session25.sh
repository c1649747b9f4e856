#!/bin/bash
# Round-3 session 25: wino4_f32 persistent scheduling: whole tiles strided (0) / contiguous (3) / split (1)
cd "$(dirname "$0")/../.."
export RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_base.so
for sp in 0 3 1; do
  echo "=== RTPOSE_W4_SPLIT=$sp"
  RTPOSE_W4_SPLIT=$sp timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E "model0|model1_1.0|^k=3|sum of"
done

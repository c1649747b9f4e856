#!/bin/bash
# Round-4 session 23: F(6,7) kernel - timing-only: segment loads of the 128-channel inputs as they would be on 8-channel planes
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for v in dev w7planes dev w7planes; do
  echo "=== $v"
  RTPOSE_LIB_PATH=$R/tools/exp/lib_$v.so python tools/profile_layers.py 32 368 368 3 fp32 2>&1 | grep -E "^model2_1|sum of|^k="
done > $O/s23_w7planes.txt 2>&1
cat $O/s23_w7planes.txt

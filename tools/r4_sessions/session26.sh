#!/bin/bash
# Round-4 session 26: left-overs of at most a quarter round: the 16 x 16 form (default) against half tiles (RTPOSE_W4_CUT_PCT=0)
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
export RTPOSE_LIB_PATH=$R/tools/exp/lib_dev.so
for cut in 25 0 25 0; do
  echo "=== RTPOSE_W4_CUT_PCT=$cut"
  RTPOSE_W4_CUT_PCT=$cut python tools/profile_layers.py 32 368 368 3 fp32 2>&1 | grep -E "^model0.2 |^model0.2[35]|^model1_1.[024]|^k=3"
done > $O/s26_cut.txt 2>&1
cat $O/s26_cut.txt

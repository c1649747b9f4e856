#!/bin/bash
# Round-4 session 17: timing-only ablations of the bf16 conv kernel with the transposed-product epilogue
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for v in dev bnostage bnear bnob bnoa bnostore bnone; do
  echo "=== $v"
  RTPOSE_LIB_PATH=$R/tools/exp/lib_$v.so python tools/profile_layers.py 32 368 368 3 bf16 2>&1 | grep -E "model0.2 |model0.21 |model2_1.2\+|sum of|^k="
done > $O/s17_ablations.txt 2>&1
cat $O/s17_ablations.txt

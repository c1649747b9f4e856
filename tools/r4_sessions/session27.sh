#!/bin/bash
# Round-4 session 27: knobs of the F(4x4,3x3) kernel re-measured on channel planes (patch-load position, priorities, A ring)
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for v in dev w4_l0 w4_l3 w4_prio2 w4_prio3 w4_a3 dev; do
  echo "=== $v"
  RTPOSE_LIB_PATH=$R/tools/exp/lib_$v.so python tools/profile_layers.py 32 368 368 3 fp32 2>&1 | grep -E "^model0.2 |^model0.12|^model0.21|^k=3"
done > $O/s27_knobs.txt 2>&1
cat $O/s27_knobs.txt

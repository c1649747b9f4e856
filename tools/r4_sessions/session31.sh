#!/bin/bash
# Round-4 session 31: channel planes between the 7x7 convs, A/B on one box (developer build, RTPOSE_PLANES=1: the 3x3 chain only;
# 2: the 7x7 buffers as well), then the new parity tests
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for lvl in 1 2 1 2 1 2; do
  echo "=== RTPOSE_PLANES=$lvl"
  RTPOSE_LIB_PATH=$R/tools/exp/lib_dev.so RTPOSE_PLANES=$lvl python tools/profile_layers.py 32 368 368 3 fp32 2>&1 | grep -E "^model3_1.[02468]|sum of|^k=7"
done > $O/s31_w7planes_ab.txt 2>&1
cat $O/s31_w7planes_ab.txt
( timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q -x --timeout 900 -k "planes" 2>&1 | tail -5 ) > $O/s31_tests.log 2>&1
cat $O/s31_tests.log

#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_net_gpu.py -m gpu -q --timeout 600 2>&1 | tail -30 ) > gpurun_out/s5_tests.log 2>&1
for fs in 2 1; do
  echo "=== RTPOSE_W7_FS=$fs"
  RTPOSE_W7_FS=$fs RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_base.so timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E "model2_1.0|model2_1.2|model3_1.4|^k=|sum of"
done > gpurun_out/s5_fs.log 2>&1
tail -n 6 gpurun_out/s5_tests.log; cat gpurun_out/s5_fs.log

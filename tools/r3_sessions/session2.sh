#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_net_gpu.py -m gpu -q --timeout 600 2>&1 | tail -40 ) > gpurun_out/s2_conv_tests.log 2>&1
( VARIANTS="base:: pfd2:: tl3::" timeout 600 tools/r3_variants.sh run ) > gpurun_out/s2_variants.log 2>&1
( RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_tl3.so RTPOSE_TIMELINE_WM=1 timeout 200 python tools/timeline_w3.py ) > gpurun_out/s2_timeline_wm1.log 2>&1
( RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_tl3.so RTPOSE_TIMELINE_WM=2 timeout 200 python tools/timeline_w3.py ) > gpurun_out/s2_timeline_wm2.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_runtime_gpu.py --deselect tests/test_conv_gpu.py --deselect tests/test_net_gpu.py 2>&1 | tail -60 ) > gpurun_out/s2_tests.log 2>&1
tail -n 3 gpurun_out/s2_conv_tests.log gpurun_out/s2_tests.log
cat gpurun_out/s2_variants.log gpurun_out/s2_timeline_wm1.log gpurun_out/s2_timeline_wm2.log | grep -v amdgpu.ids

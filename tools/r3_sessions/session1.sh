#!/bin/bash
# GPU session 1 of round 3: A/B of the Winograd kernel variants, per-tile timeline of the 3x3 kernel, full GPU tests
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 tools/r3_variants.sh run ) > gpurun_out/s1_variants.log 2>&1
( RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_tl3.so RTPOSE_TIMELINE_WM=1 timeout 200 python tools/timeline_w3.py ) > gpurun_out/s1_timeline_wm1.log 2>&1
( RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_tl3.so RTPOSE_TIMELINE_WM=2 timeout 200 python tools/timeline_w3.py ) > gpurun_out/s1_timeline_wm2.log 2>&1
( timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --deselect tests/test_runtime_gpu.py 2>&1 | tail -150 ) > gpurun_out/s1_tests.log 2>&1
( timeout 900 python -m pytest tests/test_runtime_gpu.py -m gpu -q -s 2>&1 | tail -60 ) > gpurun_out/s1_runtime_tests.log 2>&1
tail -3 gpurun_out/s1_tests.log gpurun_out/s1_runtime_tests.log
cat gpurun_out/s1_variants.log | tail -60

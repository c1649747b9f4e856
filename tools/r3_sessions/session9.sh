#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { echo "=== $1"; RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$1.so timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E "model2_1.0|model2_1.2|^k=7|sum of"; }
( run base; run xsplit; run base; run xsplit ) > gpurun_out/s9_xsplit.log 2>&1
RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_xsplit.so timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -k "winograd7" 2>&1 | tail -3 >> gpurun_out/s9_xsplit.log
cat gpurun_out/s9_xsplit.log

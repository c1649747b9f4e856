#!/bin/bash
# Round-4 session 10: stem conv on the matrix pipe
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_shufflenet_gpu.py -q --timeout 800 2>&1 | tail -12 ) > gpurun_out/s10_tests.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s10_sn_fp32.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 bf16 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s10_sn_bf16.log 2>&1
tail -n 5 gpurun_out/s10_tests.log; grep -E "stage1|launches" gpurun_out/s10_sn_fp32.log gpurun_out/s10_sn_bf16.log

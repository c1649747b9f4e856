#!/bin/bash
# Round-4 session 9: zero-copy channel shuffle in both ShuffleNetV2 plans (shared slot allocator)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_shufflenet_gpu.py tests/test_pw_fused_gpu.py -q --timeout 800 2>&1 | tail -30 ) > gpurun_out/s9_tests.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s9_sn_fp32.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 bf16 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s9_sn_bf16.log 2>&1
tail -n 30 gpurun_out/s9_tests.log; tail -n 8 gpurun_out/s9_sn_fp32.log; cat gpurun_out/s9_sn_bf16.log

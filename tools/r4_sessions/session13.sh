#!/bin/bash
# Round-4 session 13: magic-number division in the pointwise kernels
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_shufflenet_gpu.py tests/test_pw_fused_gpu.py -q --timeout 800 2>&1 | tail -12 ) > gpurun_out/s13_tests.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s13_sn_fp32.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 bf16 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s13_sn_bf16.log 2>&1
tail -n 4 gpurun_out/s13_tests.log; grep -E "network.3.1|network.4.1|network.5.0|network.5.1|conv5|^pw|^fused|launches" gpurun_out/s13_sn_fp32.log; grep -E "network.4.1|network.5.1|conv5|paf|^pw|^fused|launches" gpurun_out/s13_sn_bf16.log

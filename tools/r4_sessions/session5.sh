#!/bin/bash
# Round-4 session 5: wave-autonomous transposed pointwise kernels (pw_t.hip) in the fp32 ShuffleNetV2 plan
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_pw_fused_gpu.py -q -x --timeout 800 2>&1 | tail -25 ) > gpurun_out/s5_tests_pw.log 2>&1
( timeout 900 python -m pytest tests/test_shufflenet_gpu.py -q --timeout 800 2>&1 | tail -25 ) > gpurun_out/s5_tests_sn.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s5_shufflenet_fp32.log 2>&1
tail -n 15 gpurun_out/s5_tests_pw.log; tail -n 15 gpurun_out/s5_tests_sn.log; tail -n 50 gpurun_out/s5_shufflenet_fp32.log

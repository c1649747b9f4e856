#!/bin/bash
# Round-4 session 3: head kernel parity (test fixed), shufflenet suite, host-copy probe
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_pw_fused_gpu.py tests/test_shufflenet_gpu.py -q --timeout 800 2>&1 | tail -25 ) > gpurun_out/s3_tests.log 2>&1
( timeout 300 python tools/exp/host_copy_probe.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s3_probe.log 2>&1
tail -n 12 gpurun_out/s3_tests.log; cat gpurun_out/s3_probe.log

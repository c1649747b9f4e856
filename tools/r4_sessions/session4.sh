#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout 400 python tools/bench_streaming.py 12 32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s4_streaming.log 2>&1
( timeout 600 python -m pytest tests/test_dropin_gpu.py -q -x --timeout 500 -k "streaming" 2>&1 | tail -5 ) > gpurun_out/s4_tests.log 2>&1
cat gpurun_out/s4_streaming.log gpurun_out/s4_tests.log

#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_net_gpu.py tests/test_dropin_gpu.py -m gpu -q --timeout 600 2>&1 | tail -30 ) > gpurun_out/s4_tests.log 2>&1
( timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 ) > gpurun_out/s4_layers.log 2>&1
tail -n 4 gpurun_out/s4_tests.log; grep -E "nchw|model0.0 |model0.2 |^k=|sum of" gpurun_out/s4_layers.log

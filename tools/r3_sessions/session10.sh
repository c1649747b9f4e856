#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_net_gpu.py -m gpu -q --timeout 600 2>&1 | tail -15 ) > gpurun_out/s10_tests.log 2>&1
cat gpurun_out/s10_tests.log

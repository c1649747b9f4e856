#!/bin/bash
# first GPU contact of the bf16 path: parity tests, then per-launch timings
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -25 > gpurun_out/bf16_tests.log
cat gpurun_out/bf16_tests.log
timeout 300 python tools/profile_layers.py 32 368 368 3 bf16 > gpurun_out/bf16_layers.log 2>&1
tail -60 gpurun_out/bf16_layers.log

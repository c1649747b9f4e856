#!/bin/bash
# Round-3 session 24: F(4x4,3x3) with split tiles - parity + timing
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_conv_gpu.py -q -x -k "winograd4" 2>&1 | tail -15
timeout 600 python tools/bench_conv3.py 2>&1 | grep -v amdgpu.ids | sed -E 's/direct [0-9.]+ ms \([0-9]+ TF\/s\)  //'
timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E "model0|model1_1|^k=|sum of"

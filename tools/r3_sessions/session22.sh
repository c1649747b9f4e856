#!/bin/bash
# Round-3 session 22: F(4x4,3x3) as the plan default - whole-network tests, per-layer profile, bench
cd "$(dirname "$0")/../.."
timeout 1500 python -m pytest tests/test_net_gpu.py tests/test_wino_numerics_gpu.py tests/test_conv_gpu.py -q -m gpu 2>&1 | tail -30
timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E "model0|model1_1|^k=|sum of"
timeout 600 python bench.py 2>&1 | tail -2

#!/bin/bash
# Round-3 session 31: segment / patch loads of the non-transforming waves through a zero-extent descriptor (7x7 8-wave form,
# F(4x4,3x3) small form): parity + timing
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_net_gpu.py -q -x 2>&1 | tail -4
timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E "model2_1.2|model2_1.0|model3_1.4|^k=|sum of"
timeout 300 python tools/latency_b1.py 2>&1 | grep fp32

#!/bin/bash
# Round-3 session 28: F(4x4,3x3): whole rounds persistent + the rest in the 16 x 16 form
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_net_gpu.py -q -x -k "winograd4 or net" 2>&1 | tail -8
timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E "model0|model1_1|^k=|sum of"
for n in 8 16 24; do echo "batch $n: $(timeout 300 python tools/profile_layers.py $n 368 368 5 fp32 2>&1 | grep -E '^k=3|sum of' | tr '\n' ' ')"; done

#!/bin/bash
# Round-3 session 13: F(4x4,3x3) first light - parity tests + per-shape timing against direct / F(2x2,3x3)
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_conv_gpu.py -q -x -k "winograd4" 2>&1 | tail -15
timeout 600 python tools/bench_conv3.py 2>&1 | grep -v amdgpu.ids

#!/bin/bash
# Round-3 session 26: small-grid form of F(4x4,3x3) - parity, batch-1 latency, small batches
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_conv_gpu.py -q -x -k "winograd4" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_net_gpu.py -q -x 2>&1 | tail -5
timeout 300 python tools/latency_b1.py 2>&1 | tail -3
for n in 1 2 4 8; do
  echo "=== per-layer batch $n"
  timeout 300 python tools/profile_layers.py $n 368 368 5 fp32 2>&1 | grep -E "^k=3|sum of"
done

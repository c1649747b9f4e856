#!/bin/bash
# Round-3 session 23: F(4x4,3x3) default - numerics tests, batch-1 latency (F(4x4) vs F(2x2) 3x3 forms), small batches
cd "$(dirname "$0")/../.."
timeout 1500 python -m pytest tests/test_wino_numerics_gpu.py -q -m gpu -s 2>&1 | grep -E "passed|failed|FAILED|amplification|Error|assert" | tail -20
echo "=== latency b1, default (F(4x4,3x3))"
timeout 300 python tools/latency_b1.py 2>&1 | tail -4
echo "=== latency b1, RTPOSE_WINOGRAD3_M=2 (F(2x2,3x3))"
RTPOSE_WINOGRAD3_M=2 timeout 300 python tools/latency_b1.py 2>&1 | tail -4
for n in 1 4 8; do
  echo "=== per-layer batch $n default"
  timeout 300 python tools/profile_layers.py $n 368 368 5 fp32 2>&1 | grep -E "^k=3|sum of"
  echo "=== per-layer batch $n F(2x2,3x3)"
  RTPOSE_WINOGRAD3_M=2 timeout 300 python tools/profile_layers.py $n 368 368 5 fp32 2>&1 | grep -E "^k=3|sum of"
done

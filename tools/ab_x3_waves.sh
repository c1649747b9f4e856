#!/bin/bash
[ -f "$(dirname "$0")/exp/lib_dev.so" ] || "$(dirname "$0")/build_dev.sh"
export RTPOSE_LIB_PATH="$(cd "$(dirname "$0")" && pwd)/exp/lib_dev.so"   # env knobs exist in developer builds only
cd "$(dirname "$0")/.."
python -m pytest tests/test_bf16x3_gpu.py -x -q 2>&1 | tail -2
for w in 14 22 41; do
  echo "=== RTPOSE_BF16_WAVES=$w"
  RTPOSE_BF16_WAVES=$w python tools/profile_layers.py 32 368 368 3 bf16x3 2>&1 | grep -E "model0.21 |model2_1.2\+|sum of|^k=[37]"
done

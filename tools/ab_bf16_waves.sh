#!/bin/bash
[ -f "$(dirname "$0")/exp/lib_dev.so" ] || "$(dirname "$0")/build_dev.sh"
export RTPOSE_LIB_PATH="$(cd "$(dirname "$0")" && pwd)/exp/lib_dev.so"   # env knobs exist in developer builds only
cd "$(dirname "$0")/.."
python -m pytest tests/test_bf16_gpu.py -x -q 2>&1 | tail -3
for w in 14 22; do
  echo "=== RTPOSE_BF16_WAVES=$w"
  RTPOSE_BF16_WAVES=$w python tools/profile_layers.py 32 368 368 3 bf16 2>&1 | grep -E "model0.2 |model0.7 |model0.21 |model2_1.2\+|sum of|^k="
done

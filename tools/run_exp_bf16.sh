#!/bin/bash
cd "$(dirname "$0")/.."
echo "=== product"; python tools/profile_layers.py 32 368 368 3 bf16 2>&1 | grep -E "model0.2 |model0.21 |model2_1.2\+|sum of|^k="
for v in "$@"; do
  echo "=== $v"
  RTPOSE_LIB_PATH=$PWD/tools/exp/lib_$v.so python tools/profile_layers.py 32 368 368 3 bf16 2>&1 | grep -E "model0.2 |model0.21 |model2_1.2\+|sum of|^k="
done

#!/bin/bash
# runs tools/profile_layers.py against each experimental library variant
cd "$(dirname "$0")/.."
for v in "$@"; do
  echo "=== $v"
  RTPOSE_LIB_PATH=$PWD/tools/exp/lib_$v.so python tools/profile_layers.py 32 368 368 3 2>&1 | grep -E "model0.2 |model0.21 |model2_1.0\+|model2_1.2\+|sum of|^k="
done

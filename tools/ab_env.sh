#!/bin/bash
# usage: tools/ab_env.sh VAR v1 v2 ...   (bf16 per-launch profile under each value of an env switch)
[ -f "$(dirname "$0")/exp/lib_dev.so" ] || "$(dirname "$0")/build_dev.sh"
export RTPOSE_LIB_PATH="$(cd "$(dirname "$0")" && pwd)/exp/lib_dev.so"   # env knobs exist in developer builds only
cd "$(dirname "$0")/.."
VAR=$1; shift
for v in "$@"; do
  echo "=== $VAR=$v"
  env $VAR=$v python tools/profile_layers.py 32 368 368 3 bf16 2>&1 | grep -E "model0.2 |model0.21 |model1_1.0|model2_1.0\+|model2_1.2\+|sum of|^k=[37]"
done

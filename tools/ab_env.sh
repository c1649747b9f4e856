#!/bin/bash
# usage: tools/ab_env.sh VAR v1 v2 ...   (bf16 per-launch profile under each value of an env switch)
cd "$(dirname "$0")/.."
VAR=$1; shift
for v in "$@"; do
  echo "=== $VAR=$v"
  env $VAR=$v python tools/profile_layers.py 32 368 368 3 bf16 2>&1 | grep -E "model0.2 |model0.21 |model1_1.0|model2_1.0\+|model2_1.2\+|sum of|^k=[37]"
done

#!/bin/bash
# usage: tools/ab_env_f32.sh VAR v1 v2 ...   (fp32 per-launch profile under each value of an env switch)
[ -f "$(dirname "$0")/exp/lib_dev.so" ] || "$(dirname "$0")/build_dev.sh"
export RTPOSE_LIB_PATH="$(cd "$(dirname "$0")" && pwd)/exp/lib_dev.so"   # env knobs exist in developer builds only
cd "$(dirname "$0")/.."
VAR=$1; shift
for v in "$@"; do
  echo "=== $VAR=$v"
  env $VAR=$v python tools/profile_layers.py 32 368 368 3 2>&1 | grep -E "model0.5 |model0.10 |model0.12 |model0.14 |sum of|^k=[37]"
done

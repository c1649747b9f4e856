#!/bin/bash
# Round-3 session 38: F(4x4,3x3) as it is now: what the filter stream costs (timing-only ablation)
cd "$(dirname "$0")/../.."
for v in base w4nb; do
  echo "=== $v: $(RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$v.so timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E 'model0.2 |model0.7 |model0.12|model0.21|model0.23|^k=3' | awk '{print $1, $3, $4}' | tr '\n' ' ')"
done

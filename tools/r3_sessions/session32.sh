#!/bin/bash
# Round-3 session 32: F(4x4,3x3) after the vmcnt fix: load position, and what the loads / the transform still cost
cd "$(dirname "$0")/../.."
for v in base w4l2 w4l4 w4nl w4ns; do
  echo "=== $v: $(RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$v.so timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E 'model0.2 |model0.12|model0.21|^k=3' | awk '{print $1, $4, $5}' | tr '\n' ' ')"
done

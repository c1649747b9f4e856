#!/bin/bash
# Round-3 session 34: F(4x4,3x3): how large a left-over round is worth cutting off into the small form
cd "$(dirname "$0")/../.."
export RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_base.so
for c in 0 25 30 50; do
  echo "=== RTPOSE_W4_CUT_PCT=$c: $(RTPOSE_W4_CUT_PCT=$c timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E 'model0.12|model0.19|model0.21|model0.23|^k=3' | awk '{print $1, $3, $4}' | tr '\n' ' ')"
done
for n in 8 16 24; do for c in 25 50; do echo "batch $n cut $c: $(RTPOSE_W4_CUT_PCT=$c timeout 300 python tools/profile_layers.py $n 368 368 5 fp32 2>&1 | grep -E '^k=3')"; done; done

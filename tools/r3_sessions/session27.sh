#!/bin/bash
# Round-3 session 27: where the 16 x 16 form of F(4x4,3x3) stops paying (rounds of 32 x 64 tiles)
cd "$(dirname "$0")/../.."
export RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_base.so
for r in 0 1 2 4; do
  for n in 1 2 4 8 16; do
    echo "=== RTPOSE_W4_SMALL_ROUNDS=$r batch $n: $(RTPOSE_W4_SMALL_ROUNDS=$r timeout 300 python tools/profile_layers.py $n 368 368 5 fp32 2>&1 | grep -E "^k=3" )"
  done
done
echo "=== batch 1 per layer, rounds=2"
RTPOSE_W4_SMALL_ROUNDS=2 timeout 300 python tools/profile_layers.py 1 368 368 5 fp32 2>&1 | grep -E "k=3 "

#!/bin/bash
# Round-3 session 39: per-chunk timeline of wino4_f32 (256 -> 256 layer and conv1_2)
cd "$(dirname "$0")/../.."
export RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_tl4.so
echo "=== 256 -> 256 (conv3_2..4)"; timeout 200 python tools/timeline_w4.py 2>&1 | grep -v amdgpu
echo "=== 64 -> 64 (conv1_2)"; RTPOSE_TIMELINE_W4=64,64 timeout 200 python tools/timeline_w4.py 2>&1 | grep -v amdgpu

#!/bin/bash
# Round-3 session 40: per-chunk timeline of wino4_f32 with s_setprio schemes for the two waves of a SIMD
cd "$(dirname "$0")/../.."
for v in ${VLIST:-tl4p1 tl4p2 tl4p3}; do
  echo "=== $v"; RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$v.so timeout 200 python tools/timeline_w4.py 2>&1 | grep -E "period|wave [0-9]|epilogue|total"
done

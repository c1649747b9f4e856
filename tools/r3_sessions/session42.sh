#!/bin/bash
# Round-3 session 42: wino4_f32 with the A fragments requested 8 MFMAs ahead (ring of three): parity, timing, timeline
cd "$(dirname "$0")/../.."
RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_a3.so timeout 300 python -m pytest tests/test_conv_gpu.py -q -x -k "winograd4" 2>&1 | tail -2
for v in base a3; do
  echo "=== $v: $(RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$v.so timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E 'model0.2 |model0.12|model0.21|^k=3|sum of' | awk '{print $1, $3, $4, $5}' | tr '\n' ' ')"
done
echo "=== timeline a3"; RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_tl4a3.so timeout 200 python tools/timeline_w4.py 2>&1 | grep -E "period|wave [0-9]|epilogue|total"

#!/bin/bash
# Round-3 session 11: the two-waves-per-SIMD 3x3 Winograd form (conv_wino16.hip) behind RTPOSE_W3_16 (developer builds)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_base.so
echo "=== conv tests with RTPOSE_W3_16=1 (bit-identity tests against the small-grid form are expected to differ)"
RTPOSE_W3_16=1 timeout 600 python -m pytest tests/test_conv_gpu.py -q -k "wino or Wino" 2>&1 | tail -15
SHOW="model0.5 |model0.7 |model0.10|model0.12|model0.19|model0.21|model1_1.0|^k=|sum of"
for v in base w16b w16c; do
  for sw in 0 1; do
    [ $v != base ] && [ $sw = 0 ] && continue
    echo "=== $v RTPOSE_W3_16=$sw"
    RTPOSE_W3_16=$sw RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$v.so timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E "$SHOW"
  done
done

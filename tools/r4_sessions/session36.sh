#!/bin/bash
# Round-4 session 36: F(4x4,3x3) - both transform roles on the OLDER sibling waves: parity, per-wave timeline, per-layer events, bench
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_net_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -3 ) > $O/s36_tests.log 2>&1
cat $O/s36_tests.log
RTPOSE_LIB_PATH=$R/tools/exp/lib_tl4.so python tools/timeline_w4.py 2>&1 | grep -v amdgpu > $O/s36_timeline_w4.txt
cat $O/s36_timeline_w4.txt
python tools/profile_layers.py 32 368 368 3 fp32 2>&1 | grep -E "^model0|^model1_1.[024]|sum of|^k=" > $O/s36_layers.txt
cat $O/s36_layers.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | grep -o '"value": [0-9.]*, "unit": "images/s"'

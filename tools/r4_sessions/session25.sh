#!/bin/bash
# Round-4 session 25: half tiles for the left-over round of the F(4x4,3x3) kernel - parity, then per-layer events and the bench
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_net_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -25 ) > $O/s25_tests.log 2>&1
cat $O/s25_tests.log
python tools/profile_layers.py 32 368 368 3 fp32 2>&1 | grep -E "^model0|^model1_1.[024]|sum of|^k=" > $O/s25_layers.txt
cat $O/s25_layers.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic > $O/s25_bench.json 2> $O/s25_bench.err
grep -o '"value": [0-9.]*, "unit": "images/s"' $O/s25_bench.json; grep -o '"roofline_3x3": {[^}]*}' $O/s25_bench.json

#!/bin/bash
# Round-4 session 30: channel planes between the 7x7 convs of a stage as well - parity, per-layer events, bench
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_wino_numerics_gpu.py tests/test_net_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -25 ) > $O/s30_tests.log 2>&1
cat $O/s30_tests.log
python tools/profile_layers.py 32 368 368 3 fp32 2>&1 | grep -E "^model2_1|^model6_1|sum of|^k=" > $O/s30_layers.txt
cat $O/s30_layers.txt
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic > $O/s30_bench.json 2> $O/s30_bench.err
grep -o '"value": [0-9.]*, "unit": "images/s"' $O/s30_bench.json; grep -o '"executed_frac": [0-9.]*' $O/s30_bench.json

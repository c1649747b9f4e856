#!/bin/bash
# Round-4 session 22: channel-plane activations between the F(4x4,3x3) convs - parity tests, then the bench
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_conv_gpu.py tests/test_wino_numerics_gpu.py tests/test_net_gpu.py tests/test_dropin_gpu.py -m gpu -q -x --timeout 900 2>&1 | tail -25 ) > $O/s22_tests.log 2>&1
cat $O/s22_tests.log
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/s22_bench.json 2> $O/s22_bench.err
cat $O/s22_bench.json
python tools/profile_layers.py 32 368 368 3 fp32 2>&1 | grep -E "^model0|^model1_1.[024]|sum of|^k=" > $O/s22_layers.txt
cat $O/s22_layers.txt

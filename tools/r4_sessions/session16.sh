#!/bin/bash
# Round-4 session 16: transposed-product epilogue of the bf16 conv kernel - parity tests, then per-layer events with the
# production library (TR) and the developer build with RTPOSE_BF16_TR=0 (the LDS-slab epilogue)
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_bf16_gpu.py tests/test_bf16x3_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -15 ) > $O/s16_tests.log 2>&1
cat $O/s16_tests.log
for dt in bf16 bf16x3; do
  python tools/profile_layers.py 32 368 368 5 $dt 2>&1 | grep -v amdgpu.ids > $O/s16_layers_${dt}_tr.txt
  RTPOSE_LIB_PATH=$R/tools/exp/lib_dev.so RTPOSE_BF16_TR=0 python tools/profile_layers.py 32 368 368 5 $dt 2>&1 | grep -v amdgpu.ids > $O/s16_layers_${dt}_slab.txt
  tail -6 $O/s16_layers_${dt}_tr.txt; tail -6 $O/s16_layers_${dt}_slab.txt
done

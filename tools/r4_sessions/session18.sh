#!/bin/bash
# Round-4 session 18: per-block stamps of the last 7x7 bf16 launch (transposed-product epilogue)
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
RTPOSE_LIB_PATH=$R/tools/exp/lib_btime.so python tools/timeline_bf16.py 2>&1 | grep -v amdgpu.ids > $O/s18_timeline_tr.txt
RTPOSE_LIB_PATH=$R/tools/exp/lib_btime.so RTPOSE_BF16_TR=0 python tools/timeline_bf16.py 2>&1 | grep -v amdgpu.ids > $O/s18_timeline_slab.txt
cat $O/s18_timeline_tr.txt $O/s18_timeline_slab.txt

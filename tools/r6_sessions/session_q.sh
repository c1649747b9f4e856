#!/bin/bash
# Round 6, session q: conv_c64_bf16 after the epilogue / ring / load-pipelining changes: its tests, the phase timeline, the layers.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6q
mkdir -p $OUT
( timeout 600 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "conv" ) > $OUT/gpu_tests_bf16_conv.txt 2>&1
tail -3 $OUT/gpu_tests_bf16_conv.txt | tee -a $OUT/summary.txt
RTPOSE_LIB_PATH=tools/exp/lib_c64tl.so timeout 300 python tools/exp/c64_timeline.py 1 64 2>&1 | grep -v amdgpu.ids > $OUT/timeline_pool64.txt
head -14 $OUT/timeline_pool64.txt | tee -a $OUT/summary.txt; tail -1 $OUT/timeline_pool64.txt | tee -a $OUT/summary.txt
RTPOSE_LIB_PATH=tools/exp/lib_c64tl.so timeout 300 python tools/exp/c64_timeline.py 0 128 2>&1 | grep -v amdgpu.ids > $OUT/timeline_128.txt
head -1 $OUT/timeline_128.txt | tee -a $OUT/summary.txt; tail -1 $OUT/timeline_128.txt | tee -a $OUT/summary.txt
timeout 600 python tools/profile_layers.py 32 368 368 3 bf16 > $OUT/layers.txt 2>&1
grep -E "model0\.(0|2|5|7) |sum of launches|k=3" $OUT/layers.txt | tee -a $OUT/summary.txt

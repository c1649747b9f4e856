#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { echo "=== $1 FS=$2"; RTPOSE_W7_FS=$2 RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$1.so timeout 300 python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E "model2_1.0|model2_1.2|^k=7|sum of"; }
( run base 2; run pf6 2; run sets3 2; run base 1 ) > gpurun_out/s6_fs.log 2>&1
cat gpurun_out/s6_fs.log

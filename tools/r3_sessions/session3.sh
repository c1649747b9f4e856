#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in base pfd2 noasm; do
  echo "=== $v"
  RTPOSE_LIB_PATH=$PWD/tools/exp/lib_r3_$v.so timeout 300 python tools/exp/dbg_w3.py 2>&1 | grep -v amdgpu.ids
done > gpurun_out/s3_dbg.log 2>&1
cat gpurun_out/s3_dbg.log

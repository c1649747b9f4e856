#!/bin/bash
# Round-4 session 7: pw_t.hip with two waves per SIMD and <= 128-column work items vs the wide one-wave form vs pw_fused.hip
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_pw_fused_gpu.py tests/test_shufflenet_gpu.py -q -x --timeout 800 2>&1 | tail -8 ) > gpurun_out/s7_tests.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s7_sn_t2.log 2>&1
( RTPOSE_LIB_PATH=$PWD/tools/exp/lib_dev.so RTPOSE_SN_PW=plain VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s7_sn_t2_plain.log 2>&1
( RTPOSE_LIB_PATH=$PWD/tools/exp/lib_dev.so RTPOSE_SN_PW=old VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s7_sn_old.log 2>&1
tail -n 3 gpurun_out/s7_tests.log
for v in t2 t2_plain old; do echo "== $v"; grep -E "network.5.0|network.5.1|network.4.1|network.3.1|^fused|^pw |conv5|launches" gpurun_out/s7_sn_$v.log; done

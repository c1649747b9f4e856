#!/bin/bash
# Round-4 session 6: pw_t.hip after the LDS-bias / batched pass-through changes; A/B against the block-cooperative form
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_pw_fused_gpu.py tests/test_shufflenet_gpu.py -q -x --timeout 800 2>&1 | tail -8 ) > gpurun_out/s6_tests.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s6_sn_all_t.log 2>&1
for v in old plain dw; do
  ( RTPOSE_LIB_PATH=$PWD/tools/exp/lib_dev.so RTPOSE_SN_PW=$v VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s6_sn_$v.log 2>&1
done
tail -n 4 gpurun_out/s6_tests.log; grep -E "network.5.1|network.4.1|network.3.1|network.5.0|conv5|^fused|^pw|launches" gpurun_out/s6_sn_all_t.log
for v in old plain dw; do echo "== $v"; grep -E "network.5.1|network.4.1|^fused|^pw |launches" gpurun_out/s6_sn_$v.log; done

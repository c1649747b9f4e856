#!/bin/bash
# Round-4 session 2: conv5 + heads as one launch (pw_head.hip): parity, ShuffleNetV2 per-launch times,
# SQ counters of the pointwise kernels (where do the waves wait?)
cd "$(dirname "$0")/../.."
R=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_pw_fused_gpu.py tests/test_shufflenet_gpu.py -q -x --timeout 800 2>&1 | tail -25 ) > gpurun_out/s2_tests.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s2_shufflenet_fp32.log 2>&1
( VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 bf16 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s2_shufflenet_bf16.log 2>&1
( timeout 600 python -m pytest tests/test_runtime_gpu.py tests/test_dropin_gpu.py -q -x --timeout 500 -k "c_host or streaming" 2>&1 | tail -15 ) > gpurun_out/s2_tests_host.log 2>&1
( timeout 400 python tools/bench_streaming.py 12 32 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s2_streaming.log 2>&1
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES \
  --kernel-trace -d $R/gpurun_out/s2_pmc -o t -- python $R/tools/bench_shufflenet.py 128 2 fp32 > $R/gpurun_out/s2_pmc.log 2>&1
db=$(find $R/gpurun_out/s2_pmc -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $R/gpurun_out/s2_pmc.summary.txt 2>&1
rm -rf $R/gpurun_out/s2_pmc
cd $R
tail -n 8 gpurun_out/s2_tests.log; tail -n 8 gpurun_out/s2_tests_host.log; cat gpurun_out/s2_streaming.log; tail -n 50 gpurun_out/s2_shufflenet_fp32.log; tail -n 6 gpurun_out/s2_shufflenet_bf16.log; head -60 gpurun_out/s2_pmc.summary.txt

#!/bin/bash
# round 5, after the evidence run: stem conv + pool (float4 rows, mask-free write-out) and the streaming / config-5 callers
# with the decoder on the side stream
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_shufflenet_gpu.py -m gpu -q -x > gpurun_out/sa_tests_sn.log 2>&1; tail -3 gpurun_out/sa_tests_sn.log
timeout 600 python -m pytest tests/test_dropin_gpu.py tests/test_runtime_gpu.py -m gpu -q -x -k "streaming or decoder_of_one_batch or two_processes or config5 or config_5" > gpurun_out/sa_tests_stream.log 2>&1; tail -3 gpurun_out/sa_tests_stream.log
VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 bf16 > gpurun_out/sa_sn_bf16.log 2>&1; tail -8 gpurun_out/sa_sn_bf16.log
VERBOSE=1 timeout 300 python tools/bench_shufflenet.py 128 10 fp32 > gpurun_out/sa_sn_fp32.log 2>&1; tail -3 gpurun_out/sa_sn_fp32.log
timeout 300 python tools/bench_streaming.py 12 32 > gpurun_out/sa_streaming.log 2>&1; cat gpurun_out/sa_streaming.log | grep host-to-host
timeout 300 python tools/bench_config5.py > gpurun_out/sa_config5.json 2> gpurun_out/sa_config5.err; cut -c1-200 gpurun_out/sa_config5.json

#!/bin/bash
# round 5: after the decoder-next-to-MFMA finding - bf16 plans guard their whole forward, limb_assign without its
# double-precision sample chain: soak, A/B against the kernel before, benches
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_runtime_gpu.py tests/test_decode_gpu.py -m gpu -q -x > gpurun_out/sb_tests.log 2>&1; tail -3 gpurun_out/sb_tests.log
timeout 600 python -m pytest tests/test_dropin_gpu.py -m gpu -q -x -k "streaming or ski or tta" > gpurun_out/sb_tests2.log 2>&1; tail -3 gpurun_out/sb_tests2.log
export SHARED_ESTIMATOR=1 SPY=none
timeout 400 python tools/exp/overlap_flake.py 40 fp32:160 bf16:320 > gpurun_out/sb_flake_new.log 2>&1; grep ": " gpurun_out/sb_flake_new.log | tail -4 | cut -c1-200
unset SHARED_ESTIMATOR SPY
python bench.py --steps 20 --warmup 3 > gpurun_out/sb_bench.json 2> gpurun_out/sb_bench.err; cut -c1-200 gpurun_out/sb_bench.json
python bench.py --steps 20 --warmup 3 --dtype bf16 > gpurun_out/sb_bench_bf16.json 2>> gpurun_out/sb_bench.err; cut -c1-200 gpurun_out/sb_bench_bf16.json
python bench.py --steps 20 --warmup 3 --dtype bf16 --decode-overlap 0 > gpurun_out/sb_bench_bf16_one.json 2>> gpurun_out/sb_bench.err; cut -c1-200 gpurun_out/sb_bench_bf16_one.json
timeout 300 python tools/bench_streaming.py 12 32 > gpurun_out/sb_streaming.log 2>&1; grep host-to-host gpurun_out/sb_streaming.log

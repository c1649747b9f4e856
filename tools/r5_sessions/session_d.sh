#!/bin/bash
# round 5, final tree: the drop-in tests that call the decoder (without the two TTA parity tests, which ran in session_b),
# smoke, the host-to-host and config-5 callers on the two-stream pipeline
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_dropin_gpu.py -m gpu -q -x -k "not tta and not streaming and not ski" > gpurun_out/sd_tests.log 2>&1; tail -2 gpurun_out/sd_tests.log
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/sd_smoke.txt 2>&1; tail -1 gpurun_out/sd_smoke.txt
timeout 120 python tools/bench_streaming.py 12 32 > gpurun_out/sd_streaming.txt 2>&1; grep host-to-host gpurun_out/sd_streaming.txt
timeout 60 python tools/bench_config5.py > gpurun_out/sd_config5.json 2> gpurun_out/sd_config5.err; cut -c1-160 gpurun_out/sd_config5.json

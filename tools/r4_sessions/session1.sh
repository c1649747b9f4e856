#!/bin/bash
# Round-4 session 1: guarded AUTO default, arena generation, F(4x4,3x3) bounds clamp, full-size hostile network,
# reduced-precision TTA against the oracle; headline bench unchanged?
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_wino_numerics_gpu.py -q -x --timeout 900 \
    -k "guarded or sibling or unknown_descriptor or clamped or unnormalised or form_is_chosen" 2>&1 | tail -25 ) > gpurun_out/s1_tests_wino.log 2>&1
( timeout 900 python -m pytest tests/test_dropin_gpu.py -q -x --timeout 800 -k "reduced_precision or multiscale" 2>&1 | tail -25 ) > gpurun_out/s1_tests_tta.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic ) > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
tail -n 12 gpurun_out/s1_tests_wino.log; tail -n 12 gpurun_out/s1_tests_tta.log; cat gpurun_out/s1_bench.json

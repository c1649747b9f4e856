#!/bin/bash
# full GPU test suite + a short bench (checkpoint run)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -60 ) > gpurun_out/check_tests.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic ) > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err
tail -n 5 gpurun_out/check_tests.log; cat gpurun_out/check_bench.json

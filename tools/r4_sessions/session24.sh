#!/bin/bash
# Round-4 session 24: the whole GPU suite on the final tree
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 ) > gpurun_out/r04_gpu_tests.txt 2>&1
cat gpurun_out/r04_gpu_tests.txt

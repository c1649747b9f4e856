#!/bin/bash
# Round-4 session 14: the whole GPU suite on the current tree
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -15 ) > gpurun_out/s14_tests.log 2>&1
cat gpurun_out/s14_tests.log

#!/bin/bash
# Round 6, session h: the full GPU suite on the final tree, then the evidence run (tools/prof_round6_final.sh).
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r06_gpu_tests.txt 2>&1
tail -5 gpurun_out/r06_gpu_tests.txt
GRAFT_REPO_ROOT=$R timeout 2400 bash tools/prof_round6_final.sh
ls -la gpurun_out/r06_* | head -50

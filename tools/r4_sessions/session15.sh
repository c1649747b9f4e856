#!/bin/bash
# Round-4 session 15: the C-host tests after the scene-file change, and the C-host evidence file on bench.py's decoder input
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_runtime_gpu.py -m gpu -q --timeout 600 2>&1 | tail -8 ) > $O/s15_tests.log 2>&1
cat $O/s15_tests.log
python - <<PY
import importlib, sys
import numpy as np
sys.path.insert(0, "$R")
synth = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd.synth")
for n in (32, 11, 8):
    heat, paf, _ = synth.make_batch(n, 368, 368, seed=100)
    with open("/tmp/scene%d.bin" % n, "wb") as f:
        f.write(np.ascontiguousarray(heat, np.float32).tobytes()); f.write(np.ascontiguousarray(paf, np.float32).tobytes())
PY
for args in "32 0 default /tmp/scene32.bin" "11 0 default /tmp/scene11.bin" "8 0 direct /tmp/scene8.bin" "32 2 default /tmp/scene32.bin" "32 1 default /tmp/scene32.bin"; do LD_LIBRARY_PATH=$R/pytorch_realtime_multi-person_pose_estimation_amd/lib timeout 300 $R/examples/c_host $args; done > $O/r04_c_host.txt 2>&1
cat $O/r04_c_host.txt

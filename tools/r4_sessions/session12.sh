#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
( timeout 300 python tools/exp/cotenant_w7.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/s12_cotenant.log 2>&1
cat gpurun_out/s12_cotenant.log

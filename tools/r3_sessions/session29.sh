#!/bin/bash
# Round-3 session 29: checks after the small-form LDS swizzle + the two-launch test; c_host with the new forms
cd "$(dirname "$0")/../.."
timeout 900 python -m pytest tests/test_conv_gpu.py -q -x -k "winograd4" 2>&1 | tail -4
timeout 300 python tools/latency_b1.py 2>&1 | grep fp32
for args in "32 0" "8 0 auto"; do LD_LIBRARY_PATH=$PWD/pytorch_realtime_multi-person_pose_estimation_amd/lib examples/c_host $args; done 2>&1 | grep -v amdgpu
cd /tmp && rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d /tmp/pmc_s29 -o t -- python $OLDPWD/tools/profile_layers.py 1 368 368 1 fp32 > /dev/null 2>&1
db=$(find /tmp/pmc_s29 -name "*.db" | head -1); [ -n "$db" ] && python $OLDPWD/tools/rocpd_summary.py $db | grep -E "wino4s.*LDS"
python $OLDPWD/tools/bench_conv3.py 2>&1 | grep -v amdgpu > $OLDPWD/gpurun_out/r03_conv3_forms.txt; tail -3 $OLDPWD/gpurun_out/r03_conv3_forms.txt

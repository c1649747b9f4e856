#!/bin/bash
# round 5: does the LDS score matrix of limb_assign hold what its lanes wrote, beside the bf16 kernels?  (probe library:
# decode.hip + a readback after the barrier, net.hip with the bf16 guard back at the last launch under RTPOSE_EXP_GUARD_LAST)
cd "$(dirname "$0")/../.."
export SHARED_ESTIMATOR=1 SPY=none RTPOSE_EXP_GUARD_LAST=1
export RTPOSE_LIB_PATH=$PWD/pytorch_realtime_multi-person_pose_estimation_amd/lib/librtpose_mi355x_probe.so
timeout 200 python tools/exp/overlap_flake.py 40 fp32:2 bf16:320 > gpurun_out/sc_probe.log 2>&1
grep -E "^(fp32|bf16): |^ +thread" gpurun_out/sc_probe.log | cut -c1-300

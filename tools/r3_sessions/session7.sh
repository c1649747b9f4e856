#!/bin/bash
# where does the bf16 TTA (BASELINE configs[2]) spend its time? kernel trace by instantiation
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/s7_tta -o t -- python $R/tools/bench_tta.py 32 3 bf16 > $O/s7_tta.log 2>&1
db=$(find $O/s7_tta -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/s7_tta_kernels.txt 2>&1
rm -rf $O/s7_tta
examples_out=$O/s7_c_host.txt
: > $examples_out
for args in "32 0" "11 0" "8 0 auto" "32 2" "32 1"; do LD_LIBRARY_PATH=$R/pytorch_realtime_multi-person_pose_estimation_amd/lib $R/examples/c_host $args >> $examples_out 2>&1; done
head -40 $O/s7_tta_kernels.txt; cat $O/s7_tta.log | grep -v amdgpu; cat $examples_out

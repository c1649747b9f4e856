#!/bin/bash
# Round-4 session 29: kernel trace of the bf16 multi-scale x4 + flip path (configs[2]): where do the 154 ms per 32 images go?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/s29_trace -o t -- python $R/tools/bench_tta.py 32 3 bf16 > $O/s29_tta.txt 2>&1
db=$(find $O/s29_trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/s29_tta_kernel_stats.txt 2>&1
rm -rf $O/s29_trace
grep -v amdgpu $O/s29_tta.txt; head -40 $O/s29_tta_kernel_stats.txt

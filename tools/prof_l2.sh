#!/bin/bash
# L2 hit-rate PMC pass (developer tool)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace -d $O/l2a -o r -- python $R/tools/profile_layers.py 32 368 368 1 > $O/l2a.log 2>&1
python $R/tools/rocpd_summary.py $O/l2a/r_results.db | grep -E "<7, 16, 0|<3, 16, 1|<3, 16, 0"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace -d $O/l2b -o r -- python $R/tools/profile_layers.py 32 368 368 1 > $O/l2b.log 2>&1
python $R/tools/rocpd_summary.py $O/l2b/r_results.db | grep -E "<7, 16, 0"
tail -3 $O/l2b.log

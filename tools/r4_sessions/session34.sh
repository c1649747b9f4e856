#!/bin/bash
# Round-4 session 34: pw_fused.hip staging lanes re-mapped against the ds_write_b128 bank conflicts - parity, events, counters
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 900 python -m pytest tests/test_pw_fused_gpu.py tests/test_shufflenet_gpu.py -m gpu -q -x --timeout 600 2>&1 | tail -4 ) > $O/s34_tests.log 2>&1
cat $O/s34_tests.log
VERBOSE=1 python tools/bench_shufflenet.py 128 10 fp32 2>&1 | grep -v amdgpu.ids | tail -8
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/s34_pmc -o t -- python $R/tools/bench_shufflenet.py 128 2 fp32 > /dev/null 2>&1
db=$(find $O/s34_pmc -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db | grep -E "pw_gemm" | cut -c1-150
rm -rf $O/s34_pmc

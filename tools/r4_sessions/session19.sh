#!/bin/bash
# Round-4 session 19: bf16 conv - non-temporal output stores and one-branch-per-XCD block order, A/B on the dev build; L2 hit counters
cd "$(dirname "$0")/../.."
R=$(pwd)
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
export RTPOSE_LIB_PATH=$R/tools/exp/lib_dev.so
run() { python tools/profile_layers.py 32 368 368 3 bf16 2>&1 | grep -E "model0.21 |model2_1.2\+|sum of|^k=[37]"; }
{
echo "=== base"; run
echo "=== NT=1"; RTPOSE_BF16_NT=1 run
echo "=== XCDGRP=1"; RTPOSE_BF16_XCDGRP=1 run
echo "=== NT=1 XCDGRP=1"; RTPOSE_BF16_NT=1 RTPOSE_BF16_XCDGRP=1 run
echo "=== base again"; run
} > $O/s19_ab.txt 2>&1
cat $O/s19_ab.txt
cd /tmp
for cfg in "0 0" "1 1"; do
  set -- $cfg
  RTPOSE_BF16_NT=$1 RTPOSE_BF16_XCDGRP=$2 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/s19_pmc -o t -- python $R/tools/profile_layers.py 32 368 368 1 bf16 > /dev/null 2>&1
  db=$(find $O/s19_pmc -name "*.db" | head -1)
  echo "=== NT=$1 XCDGRP=$2"
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db | grep -E "conv_mfma_bf16|counter" | head -40
  rm -rf $O/s19_pmc
done > $O/s19_tcc.txt 2>&1
cat $O/s19_tcc.txt

#!/bin/bash
# clock / MFMA-busy PMC pass per experimental variant
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out; mkdir -p $O
for v in "$@"; do
  RTPOSE_LIB_PATH=$R/tools/exp/lib_$v.so rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/exp_$v -o r -- python $R/tools/profile_layers.py 32 368 368 1 > $O/exp_$v.log 2>&1
  echo "== $v"; python $R/tools/rocpd_summary.py $O/exp_$v/r_results.db | grep -E "<7, 16, 0" 
done

#!/bin/bash
# Round 6, session d: packed arithmetic or re-used address registers?  (1) the real kernel, SLP build (packed instructions in
# all three): as is / with the map loads' offset registers kept live until the loads have returned / with every load landed
# before the first use; (2) the stand-alone address-register WAR victim beside the forward; (3) the full GPU suite.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6d
mkdir -p $OUT
export REPEATS=12 PEOPLE=8 RTPOSE_GUARD_OP=-1
for lib in lib_slp lib_slp_keep lib_slp_wait0 lib_slp lib_slp_keep; do
  echo "=== $lib" | tee -a $OUT/summary.txt
  RTPOSE_LIB_PATH=tools/exp/$lib.so timeout 600 python tools/exp/overlap_soak.py 4000 bf16 > $OUT/$lib.$RANDOM.log 2>&1
  grep -hE "^bf16:" $OUT/$lib.*.log | tail -1 | tee -a $OUT/summary.txt
done
unset REPEATS PEOPLE RTPOSE_GUARD_OP
echo "=== WAR victim" | tee -a $OUT/summary.txt
ONLY_WAR=1 timeout 900 python tools/exp/pk_beside_forward.py 300 bf16 none > $OUT/war.log 2>&1; echo "rc $?" >> $OUT/war.log
grep -vE "^\s*$|amdgpu.ids|SUMMARY" $OUT/war.log | tail -40 | tee -a $OUT/summary.txt
echo "=== GPU suite" | tee -a $OUT/summary.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/gpu_tests.txt 2>&1
tail -6 $OUT/gpu_tests.txt | tee -a $OUT/summary.txt

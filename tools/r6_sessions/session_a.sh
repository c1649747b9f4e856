#!/bin/bash
# Round 6, session a: which ingredient of limb_assign_kernel's sample loop (64-bit integer address arithmetic, packed-fp32
# VALU from the SLP vectoriser) or of its surroundings (stale copy of the maps, sharing CUs with the forward) is behind the
# round-5 finding (a limb score one sample off when the decoder runs beside the bf16 forward).  One box, every cell the same
# steps; the forward waits for the decoder in front of its LAST launch (RTPOSE_GUARD_OP=-1), 12 decodes per step.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6a
mkdir -p $OUT
STEPS=${STEPS:-4000}
export REPEATS=12 PEOPLE=8 RTPOSE_GUARD_OP=-1
run() {  # name, env...
  name=$1; shift
  echo "=== $name: $*" | tee -a $OUT/summary.txt
  env "$@" timeout 600 python tools/exp/overlap_soak.py $STEPS bf16 > $OUT/$name.log 2>&1
  echo "rc $?" >> $OUT/$name.log
  grep -E "^bf16:|^SUMMARY|serial path|peaks per part|Error|error" $OUT/$name.log | tee -a $OUT/summary.txt
}
run control      RTPOSE_LIB_PATH=tools/exp/lib_dev.so   RTPOSE_LIMB_A32=0
run a32          RTPOSE_LIB_PATH=tools/exp/lib_dev.so   RTPOSE_LIMB_A32=1
run noslp        RTPOSE_LIB_PATH=tools/exp/lib_noslp.so RTPOSE_LIMB_A32=0
run noslp_a32    RTPOSE_LIB_PATH=tools/exp/lib_noslp.so RTPOSE_LIMB_A32=1
run poison       RTPOSE_LIB_PATH=tools/exp/lib_dev.so   RTPOSE_LIMB_A32=0 RTPOSE_EXP_POISON=1
run cumask       RTPOSE_LIB_PATH=tools/exp/lib_dev.so   RTPOSE_LIMB_A32=0 CU_MASK=1
run control2     RTPOSE_LIB_PATH=tools/exp/lib_dev.so   RTPOSE_LIMB_A32=0
# fp32 beside its own forward (fine guard: the decoder beside the trunk), same exposure
echo "=== fp32 fine guard" | tee -a $OUT/summary.txt
env -u RTPOSE_GUARD_OP RTPOSE_GUARD_FINE=1 RTPOSE_LIB_PATH=tools/exp/lib_dev.so RTPOSE_LIMB_A32=0 timeout 600 \
  python tools/exp/overlap_soak.py 1500 fp32 > $OUT/fp32_fine.log 2>&1
grep -E "^fp32:|^SUMMARY|serial path" $OUT/fp32_fine.log | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6, session n: WHICH launches of the bf16 forward have to run beside the vectorised decoder for the finding to show.
# RTPOSE_GUARD_OP=n: the forward of step k + 1 waits for the decoder of step k in front of its launch n, so the decoder runs beside
# launches 0 .. n - 1 only (launch list of the final tree: 0 input (skipped), 1 conv1_1, 2 conv1_2, 3 conv2_1, 4 conv2_2, 5 conv3_1 ...).
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6n
mkdir -p $OUT
python tools/profile_layers.py 16 368 368 3 bf16 2>&1 | grep -v amdgpu | head -14 | tee -a $OUT/summary.txt
for n in 1 2 3 4 5 7 9 13 -1; do
  echo "=== RTPOSE_GUARD_OP=$n" | tee -a $OUT/summary.txt
  RTPOSE_LIB_PATH=tools/exp/lib_slp.so RTPOSE_GUARD_OP=$n REPEATS=12 PEOPLE=8 timeout 600 python tools/exp/overlap_soak.py 3000 bf16 > $OUT/op$n.log 2>&1
  grep -hE "^bf16:" $OUT/op$n.log | tail -1 | tee -a $OUT/summary.txt
done

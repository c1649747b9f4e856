#!/bin/bash
# Round 6, session b: (1) stand-alone packed-fp32 victim beside the library's forward; (2) are the forwards themselves
# bit-repeatable (their kernels hold packed-fp32 instructions too); (3) the production library (decoder built without the
# vectorisers) with the fine guard: long soak; (4) what the guard forms cost now; (5) same-box control with the round-5 decoder.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6b
mkdir -p $OUT
echo "=== 1 pk victim beside the forward" | tee -a $OUT/summary.txt
timeout 900 python tools/exp/pk_beside_forward.py 600 bf16 none fp32 > $OUT/pk_victim.log 2>&1; echo "rc $?" >> $OUT/pk_victim.log
grep -vE "^\s*$" $OUT/pk_victim.log | tail -40 | tee -a $OUT/summary.txt
echo "=== 2 forward determinism" | tee -a $OUT/summary.txt
timeout 900 python tools/exp/forward_determinism.py 300 > $OUT/determinism.log 2>&1; echo "rc $?" >> $OUT/determinism.log
tail -12 $OUT/determinism.log | tee -a $OUT/summary.txt
echo "=== 3 production library, fine guard, soak" | tee -a $OUT/summary.txt
RTPOSE_GUARD_FINE=1 REPEATS=12 PEOPLE=8 timeout 900 python tools/exp/overlap_soak.py 20000 bf16 > $OUT/soak_bf16.log 2>&1; echo "rc $?" >> $OUT/soak_bf16.log
grep -E "^bf16:|serial path|rc " $OUT/soak_bf16.log | tee -a $OUT/summary.txt
RTPOSE_GUARD_FINE=1 REPEATS=12 PEOPLE=8 timeout 900 python tools/exp/overlap_soak.py 3000 fp32 bf16x3 > $OUT/soak_fp32.log 2>&1; echo "rc $?" >> $OUT/soak_fp32.log
grep -E "^fp32:|^bf16x3:|serial path|rc " $OUT/soak_fp32.log | tee -a $OUT/summary.txt
echo "=== 5 same box, round-5 decoder (SLP on), guard at the last launch" | tee -a $OUT/summary.txt
RTPOSE_LIB_PATH=tools/exp/lib_slp.so RTPOSE_GUARD_OP=-1 REPEATS=12 PEOPLE=8 timeout 600 python tools/exp/overlap_soak.py 4000 bf16 > $OUT/control_slp.log 2>&1
grep -E "^bf16:|serial path" $OUT/control_slp.log | tee -a $OUT/summary.txt
echo "=== 4 bench A/B" | tee -a $OUT/summary.txt
for dt in fp32 bf16; do
  for fine in 0 1; do
    RTPOSE_GUARD_FINE=$fine timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype $dt > $OUT/bench_${dt}_fine$fine.json 2> $OUT/bench_${dt}_fine$fine.err
    echo "rc $? $dt fine=$fine: $(python -c "import json,sys; d=json.load(open('$OUT/bench_${dt}_fine$fine.json')); print(d['value'], d['ms_per_step'], d.get('records_verified'))" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
  done
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype $dt --decode-overlap 0 > $OUT/bench_${dt}_onestream.json 2>/dev/null
  echo "$dt one stream: $(python -c "import json; d=json.load(open('$OUT/bench_${dt}_onestream.json')); print(d['value'], d['ms_per_step'], d.get('records_verified'))" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done

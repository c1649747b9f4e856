#!/bin/bash
# Round 6, session u: the one-pass 1x1 pair (Mconv6 / Mconv7) as blocks of three tiles with resident filters (csrc/conv_tail.hip):
# its tests, the fp32 plan's layer times, bench.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6u
mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q -k "pair or tail or first" ) > $OUT/gpu_tests.txt 2>&1
tail -3 $OUT/gpu_tests.txt | tee -a $OUT/summary.txt
timeout 600 python tools/profile_layers.py 32 368 368 3 fp32 > $OUT/layers_fp32.txt 2>&1
grep -E "model0\.0 |k=1|sum of launches" $OUT/layers_fp32.txt | tee -a $OUT/summary.txt
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic > $OUT/bench_fp32_$i.json 2> $OUT/bench_fp32.err
python - <<PY | tee -a $OUT/summary.txt
import json
d = json.loads(open("$OUT/bench_fp32_$i.json").read().strip().splitlines()[-1])
print("bench fp32:", d["value"], d["ms_per_step"], d.get("records_verified"))
PY
done

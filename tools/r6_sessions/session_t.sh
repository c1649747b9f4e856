#!/bin/bash
# Round 6, session t: conv_first_kernel as persistent blocks (csrc/conv_first.hip): its tests, conv1_1's time in the fp32 and bf16 plans.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6t
mkdir -p $OUT
( timeout 900 python -m pytest tests -m gpu -x -q -k "first" ) > $OUT/gpu_tests_first.txt 2>&1
tail -3 $OUT/gpu_tests_first.txt | tee -a $OUT/summary.txt
for dt in fp32 bf16; do
  timeout 600 python tools/profile_layers.py 32 368 368 3 $dt > $OUT/layers_$dt.txt 2>&1
  grep -E "model0\.(0|2) |sum of launches" $OUT/layers_$dt.txt | sed "s/^/$dt: /" | tee -a $OUT/summary.txt
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err
python - <<PY | tee -a $OUT/summary.txt
import json
d = json.loads(open("$OUT/bench_fp32.json").read().strip().splitlines()[-1])
print("bench fp32:", d["value"], d["ms_per_step"], d.get("records_verified"))
PY

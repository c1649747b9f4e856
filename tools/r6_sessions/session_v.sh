#!/bin/bash
# Round 6, session v: the bf16 1x1 pair as persistent blocks (csrc/conv_tail_bf16.hip): its tests, the bf16 plan's k = 1 lines, bench.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6v
mkdir -p $OUT
( timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q ) > $OUT/gpu_tests.txt 2>&1
tail -3 $OUT/gpu_tests.txt | tee -a $OUT/summary.txt
timeout 600 python tools/profile_layers.py 32 368 368 3 bf16 > $OUT/layers_bf16.txt 2>&1
grep -E "model0\.0 |k=1|sum of launches" $OUT/layers_bf16.txt | tee -a $OUT/summary.txt
for i in 1 2; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype bf16 > $OUT/bench_bf16_$i.json 2> $OUT/bench.err
python - <<PY | tee -a $OUT/summary.txt
import json
d = json.loads(open("$OUT/bench_bf16_$i.json").read().strip().splitlines()[-1])
print("bench bf16:", d["value"], d["ms_per_step"], d.get("records_verified"))
PY
done

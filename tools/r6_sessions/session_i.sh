#!/bin/bash
# Round 6, session i: bf16 / fp32 bench lines on a fresh box first (bf16 rates fell by a third late in session h: 2780 img/s
# where session g measured 4116 and the same session's streaming run 3860), then the GPU suite with --durations.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6i
mkdir -p $OUT
for rep in 1 2; do
for args in "--dtype bf16" "--dtype fp32" "--dtype bf16 --decode-overlap 0"; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic $args > $OUT/b.json 2> $OUT/b.err
  echo "rc $? rep $rep $args: $(python -c "import json; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d.get('records_verified'))" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done
done
cp $OUT/b.json $OUT/last.json
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype bf16 > $OUT/bench_bf16.json 2>/dev/null
python tools/bench_tta.py 32 3 2>&1 | grep batched | tee -a $OUT/summary.txt
( time timeout 1500 python -m pytest tests -m gpu -q --durations=40 ) > $OUT/gpu_tests_durations.txt 2>&1
tail -60 $OUT/gpu_tests_durations.txt | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype bf16 > $OUT/bench_bf16_after_suite.json 2>/dev/null
echo "bf16 after the suite: $(python -c "import json; d=json.load(open('$OUT/bench_bf16_after_suite.json')); print(d['value'], d['ms_per_step'])")" | tee -a $OUT/summary.txt

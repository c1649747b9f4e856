#!/bin/bash
# Round 6, session j: the final tree - smoke, the driver's bench command with its wall time, the GPU suite.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6j
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt | tee -a $OUT/summary.txt
( time python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; grep real $OUT/bench_default.err | tee -a $OUT/summary.txt
python -c "
import json
d=json.loads([l for l in open('$OUT/bench_default.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['steps'], 'records_verified', d.get('records_verified'), 'frac', d['roofline']['frac'], 'executed_frac', d['roofline']['executed_frac'], 'traffic', d['roofline']['traffic'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])" | tee -a $OUT/summary.txt
( time timeout 1500 python -m pytest tests -m gpu -q ) > $OUT/gpu_tests.txt 2>&1
tail -5 $OUT/gpu_tests.txt | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic > $OUT/bench_fp32.json 2>/dev/null
python -c "
import json
d=json.load(open('$OUT/bench_fp32.json')); print('fp32', d['value'], d['ms_per_step'], d['records_verified'])" | tee -a $OUT/summary.txt
python tools/profile_layers.py 32 368 368 5 fp32 2>&1 | grep -E "k=1|\|mode|sum of" | tee -a $OUT/summary.txt

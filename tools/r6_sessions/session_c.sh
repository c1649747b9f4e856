#!/bin/bash
# Round 6, session c: (1) the stand-alone victims beside the forward (operand forms of the packed instructions; the scoring
# loop itself built with / without the SLP vectoriser); (2) the full GPU suite on the tree with the decoder rebuilt, the fine
# guard the default again and the guard carried by the plan; (3) bench lines.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6c
mkdir -p $OUT
echo "=== 1 victims beside the forward" | tee -a $OUT/summary.txt
timeout 1200 python tools/exp/pk_beside_forward.py 300 bf16 none > $OUT/victims.log 2>&1; echo "rc $?" >> $OUT/victims.log
grep -vE "^\s*$|amdgpu.ids" $OUT/victims.log | grep -v SUMMARY | tail -60 | tee -a $OUT/summary.txt
echo "=== 2 GPU suite" | tee -a $OUT/summary.txt
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/gpu_tests.txt 2>&1
tail -8 $OUT/gpu_tests.txt | tee -a $OUT/summary.txt
echo "=== 3 bench" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err; echo "rc $?" | tee -a $OUT/summary.txt
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err; echo "rc $?" | tee -a $OUT/summary.txt
python - <<'PY' | tee -a $OUT/summary.txt
import json
for n in ("fp32", "bf16"):
    try:
        d = json.load(open("gpurun_out/r6c/bench_%s.json" % n))
        print(n, d["value"], d["ms_per_step"], "records_verified", d.get("records_verified"), d["config"]["pipeline"][:60])
    except Exception as e:
        print(n, "no line:", e)
PY

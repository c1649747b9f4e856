#!/bin/bash
# Round 6, session r: evidence for the bf16 plan with conv_c64_bf16 in it: bf16 / drop-in / runtime tests, bench --dtype bf16 (twice),
# multi-scale flow (configs[2]), per-layer events, the kernel trace of the one-stream bench, batch-1 latency, streaming.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r6r
mkdir -p $OUT
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype bf16 > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err
python tools/bench_tta.py 32 3 > $OUT/tta.txt 2>&1
python tools/profile_layers.py 32 368 368 5 bf16 2>&1 | grep -v amdgpu.ids > $OUT/bf16_layers.txt
( time timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_dropin_gpu.py tests/test_runtime_gpu.py -m gpu -x -q ) > $OUT/gpu_tests.txt 2>&1
tail -5 $OUT/gpu_tests.txt | tee -a $OUT/summary.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --decode-overlap 0 --dtype bf16 > $OUT/bench_bf16_onestream.json 2> $OUT/trace.err
db=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $OUT/kernel_trace_bf16_onestream.txt 2>&1
rm -rf $OUT/trace
cd $R
python tools/latency_b1.py 2>&1 | grep batch-1 | tee -a $OUT/summary.txt
python tools/bench_streaming.py > $OUT/streaming.txt 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype bf16 > $OUT/bench_bf16_late.json 2>/dev/null
python - <<PY | tee -a $OUT/summary.txt
import json
for f in ("bench_bf16", "bench_bf16_late", "bench_bf16_onestream"):
    try:
        d = json.loads(open("$OUT/" + f + ".json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("records_verified"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -4 $OUT/tta.txt | tee -a $OUT/summary.txt
grep -E "model0\.(0|2|5|7) |sum of launches|k=3|k=7|k=1" $OUT/bf16_layers.txt | tee -a $OUT/summary.txt
grep -E "c64|conv_first|tail_bf16|conv_mfma_bf16" $OUT/kernel_trace_bf16_onestream.txt | cut -c1-150 | tee -a $OUT/summary.txt
tail -3 $OUT/streaming.txt | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6, session g: bf16 conv1_1 on the bf16 matrix instruction + the peak test with conditional neighbour loads:
# parity tests, one-stream decoder kernel times, bf16 per-layer events, bench.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r6g
mkdir -p $OUT
echo "=== tests" | tee -a $OUT/summary.txt
( time timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_net_gpu.py tests/test_decode_gpu.py tests/test_dropin_gpu.py -m gpu -x -q ) > $OUT/gpu_tests.txt 2>&1
tail -5 $OUT/gpu_tests.txt | tee -a $OUT/summary.txt
echo "=== decoder kernels, one stream" | tee -a $OUT/summary.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --decode-overlap 0 > $OUT/bench_onestream.json 2> $OUT/trace.err
db=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $OUT/kernel_trace_onestream.txt 2>&1
rm -rf $OUT/trace
grep -E "nms_refine|limb_assign|group_kernel|peak_prefix|clear_header|conv_first" $OUT/kernel_trace_onestream.txt | tee -a $OUT/summary.txt
cd $R
echo "=== bf16 layers" | tee -a $OUT/summary.txt
python tools/profile_layers.py 32 368 368 5 bf16 2>&1 | grep -v amdgpu.ids > $OUT/bf16_layers.txt
head -3 $OUT/bf16_layers.txt | tee -a $OUT/summary.txt; tail -5 $OUT/bf16_layers.txt | tee -a $OUT/summary.txt
echo "=== bench" | tee -a $OUT/summary.txt
for args in "--dtype fp32" "--dtype bf16" "--dtype bf16 --decode-overlap 0"; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic $args > $OUT/b.json 2> $OUT/b.err
  echo "rc $? $args: $(python -c "import json; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d.get('records_verified'))" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done
python tools/bench_tta.py 32 3 > $OUT/tta.txt 2>&1; grep "batched" $OUT/tta.txt | tee -a $OUT/summary.txt

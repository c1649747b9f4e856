#!/bin/bash
# Round 6, session k: grouping with the rows in registers (group_regs_kernel) and wino_amp_kernel with N_f per block: the
# decode / drop-in / runtime / numerics tests, decoder kernel times with nothing beside them, weight-load kernels, bench.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r6k
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_decode_gpu.py tests/test_dropin_gpu.py tests/test_runtime_gpu.py tests/test_wino_numerics_gpu.py -m gpu -x -q ) > $OUT/gpu_tests.txt 2>&1
tail -5 $OUT/gpu_tests.txt | tee -a $OUT/summary.txt
cd /tmp
for dt in fp32 bf16; do
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --decode-overlap 0 --dtype $dt > $OUT/bench_onestream_$dt.json 2> $OUT/trace.err
db=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $OUT/kernel_trace_onestream_$dt.txt 2>&1
rm -rf $OUT/trace
echo "--- $dt" | tee -a $OUT/summary.txt
grep -E "nms_refine|limb_assign|group_|peak_prefix|clear_header|wino_amp|wino4_amp|pack_" $OUT/kernel_trace_onestream_$dt.txt | tee -a $OUT/summary.txt
done
cd $R
for args in "--dtype fp32" "--dtype bf16" "--dtype fp32 --decode-overlap 0" "--dtype bf16 --decode-overlap 0"; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic $args > $OUT/b.json 2> $OUT/b.err
  echo "rc $? $args: $(python -c "import json; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d.get('records_verified'))" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done
python tools/latency_b1.py 2>&1 | grep batch-1 | tee -a $OUT/summary.txt

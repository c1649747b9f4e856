#!/bin/bash
# Round 6, session l: limb_assign with the one-wave greedy path: decode tests, kernel times with nothing beside them.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r6l
mkdir -p $OUT
( time timeout 1500 python -m pytest tests/test_decode_gpu.py tests/test_dropin_gpu.py tests/test_runtime_gpu.py -m gpu -x -q ) > $OUT/gpu_tests.txt 2>&1
tail -5 $OUT/gpu_tests.txt | tee -a $OUT/summary.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --decode-overlap 0 > $OUT/bench_onestream.json 2> $OUT/trace.err
db=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $OUT/kernel_trace_onestream.txt 2>&1
rm -rf $OUT/trace
grep -E "nms_refine|limb_assign|group_|clear_header" $OUT/kernel_trace_onestream.txt | tee -a $OUT/summary.txt
cd $R
python tools/latency_b1.py 2>&1 | grep batch-1 | tee -a $OUT/summary.txt

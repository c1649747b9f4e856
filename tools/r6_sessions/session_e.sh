#!/bin/bash
# Round 6, session e: (1) which group of packed instructions - the sample coordinates or the dot product - carries the finding
# (SLP build with one of them forced scalar through opaque asm); (2) the decoder after its round-6 changes (connections staged
# in LDS in limb_assign and group_kernel, peak ids written by group_kernel, batched peak-test loads): parity tests, kernel
# times from a rocprofv3 kernel trace, bench lines.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r6e
mkdir -p $OUT
export REPEATS=12 PEOPLE=8 RTPOSE_GUARD_OP=-1
for lib in lib_slp lib_slp_asm_coord lib_slp_asm_dot; do
  echo "=== $lib" | tee -a $OUT/summary.txt
  RTPOSE_LIB_PATH=tools/exp/$lib.so timeout 600 python tools/exp/overlap_soak.py 4000 bf16 > $OUT/$lib.log 2>&1
  grep -hE "^bf16:" $OUT/$lib.log | tail -1 | tee -a $OUT/summary.txt
done
unset REPEATS PEOPLE RTPOSE_GUARD_OP
echo "=== decoder tests" | tee -a $OUT/summary.txt
( time timeout 1500 python -m pytest tests/test_decode_gpu.py tests/test_dropin_gpu.py tests/test_runtime_gpu.py -m gpu -x -q ) > $OUT/gpu_tests_decoder.txt 2>&1
tail -6 $OUT/gpu_tests_decoder.txt | tee -a $OUT/summary.txt
echo "=== kernel trace" | tee -a $OUT/summary.txt
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
db=$(find $OUT/trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $OUT/bench_kernel_trace_stats.txt 2>&1
rm -rf $OUT/trace
grep -E "nms_refine|limb_assign|group_kernel|peak_prefix|clear_header|axpby" $OUT/bench_kernel_trace_stats.txt | tee -a $OUT/summary.txt
cd $R
echo "=== bench" | tee -a $OUT/summary.txt
for args in "--dtype fp32" "--dtype bf16" "--dtype fp32 --decode-overlap 0" "--dtype bf16 --decode-overlap 0"; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic $args > $OUT/b.json 2> $OUT/b.err
  echo "rc $? $args: $(python -c "import json; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d.get('records_verified'))" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done
python tools/latency_b1.py > $OUT/latency_b1.txt 2>&1; tail -8 $OUT/latency_b1.txt | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6, session f: (1) the decoder's kernels before / after the round-6 changes on ONE box, one stream (nothing beside
# them): rocprofv3 kernel trace of bench.py --decode-overlap 0; (2) the bf16 plan with its own conv1_1 kernel (conv_first.hip
# MODE 2): parity tests, per-layer events, bench; (3) decode tests on the all-limbs staging.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r6f
mkdir -p $OUT
echo "=== bf16 / net tests" | tee -a $OUT/summary.txt
( time timeout 1500 python -m pytest tests/test_bf16_gpu.py tests/test_bf16x3_gpu.py tests/test_net_gpu.py tests/test_decode_gpu.py -m gpu -x -q ) > $OUT/gpu_tests_bf16_decode.txt 2>&1
tail -5 $OUT/gpu_tests_bf16_decode.txt | tee -a $OUT/summary.txt
echo "=== decoder kernels, one stream, before / after" | tee -a $OUT/summary.txt
cd /tmp
for lib in before after; do
  if [ $lib = before ]; then export RTPOSE_LIB_PATH=$R/tools/exp/lib_before.so; else unset RTPOSE_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$lib -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --decode-overlap 0 > $OUT/bench_onestream_$lib.json 2> $OUT/trace_$lib.err
  db=$(find $OUT/trace_$lib -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $OUT/kernel_trace_onestream_$lib.txt 2>&1
  rm -rf $OUT/trace_$lib
  echo "--- $lib" | tee -a $OUT/summary.txt
  grep -E "nms_refine|limb_assign|group_kernel|peak_prefix|clear_header" $OUT/kernel_trace_onestream_$lib.txt | tee -a $OUT/summary.txt
done
unset RTPOSE_LIB_PATH
cd $R
echo "=== bf16 layers" | tee -a $OUT/summary.txt
python tools/profile_layers.py 32 368 368 5 bf16 2>&1 | grep -v amdgpu.ids > $OUT/bf16_layers.txt
head -4 $OUT/bf16_layers.txt | tee -a $OUT/summary.txt; tail -6 $OUT/bf16_layers.txt | tee -a $OUT/summary.txt
echo "=== bench" | tee -a $OUT/summary.txt
for args in "--dtype fp32" "--dtype bf16" "--dtype fp32 --decode-overlap 0" "--dtype bf16 --decode-overlap 0"; do
  timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic $args > $OUT/b.json 2> $OUT/b.err
  echo "rc $? $args: $(python -c "import json; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d.get('records_verified'))" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done
for args in "--dtype fp32 --decode-overlap 0" "--dtype bf16 --decode-overlap 0"; do
  RTPOSE_LIB_PATH=$R/tools/exp/lib_before.so timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic $args > $OUT/b.json 2> $OUT/b.err
  echo "rc $? BEFORE-lib $args: $(python -c "import json; d=json.load(open('$OUT/b.json')); print(d['value'], d['ms_per_step'], d.get('records_verified'))" 2>&1 | tail -1)" | tee -a $OUT/summary.txt
done
python tools/bench_tta.py 32 3 > $OUT/tta.txt 2>&1; tail -5 $OUT/tta.txt | tee -a $OUT/summary.txt

#!/bin/bash
# Round 6, session o: the register-resident grouping kernel (rolled form, no scratch): kernel times with nothing beside them for
# both forms on the same box (RTPOSE_GROUP_REGS=0 = the LDS form), then the decoder / drop-in / runtime GPU tests.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r6o
mkdir -p $OUT
cd /tmp
for regs in 1 0; do
  RTPOSE_GROUP_REGS=$regs rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --decode-overlap 0 > $OUT/bench_onestream_regs$regs.json 2> $OUT/trace.err
  db=$(find $OUT/trace -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $OUT/kernel_trace_onestream_regs$regs.txt 2>&1
  rm -rf $OUT/trace
  echo "=== RTPOSE_GROUP_REGS=$regs" | tee -a $OUT/summary.txt
  grep -E "nms_refine|limb_assign|group_|clear_header" $OUT/kernel_trace_onestream_regs$regs.txt | tee -a $OUT/summary.txt
  grep -o '"records_verified": [0-9]*' $OUT/bench_onestream_regs$regs.json | tee -a $OUT/summary.txt
done
cd $R
( time timeout 1500 python -m pytest tests/test_decode_gpu.py tests/test_dropin_gpu.py tests/test_runtime_gpu.py -m gpu -x -q ) > $OUT/gpu_tests.txt 2>&1
tail -5 $OUT/gpu_tests.txt | tee -a $OUT/summary.txt
python tools/latency_b1.py 2>&1 | grep batch-1 | tee -a $OUT/summary.txt

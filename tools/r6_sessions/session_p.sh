#!/bin/bash
# Round 6, session p: the 64-input-channel 3x3 bf16 kernel (csrc/conv_c64_bf16.hip): its test, the bf16 conv / plan tests, the
# bf16 plan's per-layer times with and without it (RTPOSE_BF16_C64=0 = the generic kernel) on one box, bench --dtype bf16.
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r6p
mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q ) > $OUT/gpu_tests_bf16.txt 2>&1
tail -5 $OUT/gpu_tests_bf16.txt | tee -a $OUT/summary.txt
run_layers() {  # tag, env...
  tag=$1; shift
  echo "=== layers $tag" | tee -a $OUT/summary.txt
  env "$@" timeout 600 python tools/profile_layers.py 32 368 368 3 bf16 > $OUT/layers_$tag.txt 2>&1
  grep -E "model0\.(0|2|5|7) |sum of launches|k=3" $OUT/layers_$tag.txt | tee -a $OUT/summary.txt
}
run_bench() {
  tag=$1; shift
  env "$@" timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --no-traffic > $OUT/bench_bf16_$tag.json 2> $OUT/bench_err_$tag.txt
  python - <<PY | tee -a $OUT/summary.txt
import json
try:
    d = json.loads(open("$OUT/bench_bf16_$tag.json").read().strip().splitlines()[-1])
    print("bench bf16 $tag:", d["value"], d["ms_per_step"], d.get("records_verified"))
except Exception as e:
    print("bench bf16 $tag failed:", e)
PY
}
run_layers production X=1
run_layers dev_c64_on RTPOSE_LIB_PATH=tools/exp/lib_dev.so RTPOSE_BF16_C64=1
run_layers dev_c64_off RTPOSE_LIB_PATH=tools/exp/lib_dev.so RTPOSE_BF16_C64=0
run_bench production X=1
run_bench dev_c64_off RTPOSE_LIB_PATH=tools/exp/lib_dev.so RTPOSE_BF16_C64=0
run_bench production_again X=1

# round-5 evidence: the bench line (with roofline.traffic and cpu_baseline), rocprofv3 kernel trace of the same command,
# per-launch events, PMC passes (SQ set; FETCH_SIZE; WRITE_SIZE + MFMA counts: separate passes, --kernel-trace only),
# ShuffleNetV2 (configs[3]) per-launch events + trace + FETCH / WRITE passes, and the secondary tools.
# Summaries -> gpurun_out/r05_*; copied into profiles/ by hand.  (The GPU suite: python -m pytest tests -m gpu -q > gpurun_out/r05_gpu_tests.txt.)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
python $R/bench.py --steps 20 --warmup 3 > $O/r05_bench.json 2> $O/r05_bench.err
rocprofv3 --kernel-trace --stats -d $O/r05_trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r05_bench_under_rocprof.json 2> $O/r05_trace.err
db=$(find $O/r05_trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/r05_bench_kernel_trace_stats.txt 2>&1
rm -rf $O/r05_trace
python $R/tools/profile_layers.py 32 368 368 5 fp32 > $O/r05_fp32_layers.txt 2>&1
: > $O/r05_pmc_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA"; do
  rocprofv3 --pmc $set --kernel-trace -d $O/r05_pmc -o t -- python $R/tools/profile_layers.py 32 368 368 1 fp32 > /dev/null 2>&1
  db=$(find $O/r05_pmc -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db | grep -E "wino|conv_mfma_f32|conv_first|tail_kernel|counter" >> $O/r05_pmc_counters.txt
  rm -rf $O/r05_pmc
done
# ShuffleNetV2 x1.0, 128 x 368 x 368 (configs[3])
for dt in fp32 bf16; do
  VERBOSE=1 python $R/tools/bench_shufflenet.py 128 10 $dt 2>&1 | grep -v amdgpu.ids > $O/r05_shufflenet_${dt}_events.txt
done
rocprofv3 --kernel-trace --stats -d $O/r05_sn_trace -o t -- python $R/tools/bench_shufflenet.py 128 5 fp32 > /dev/null 2>&1
db=$(find $O/r05_sn_trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/r05_shufflenet_kernel_trace_stats.txt 2>&1
rm -rf $O/r05_sn_trace
: > $O/r05_shufflenet_pmc_fetch_write.txt
for dt in fp32 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $O/r05_sn_pmc -o t -- python $R/tools/bench_shufflenet.py 128 2 $dt > /dev/null 2>&1
    db=$(find $O/r05_sn_pmc -name "*.db" | head -1)
    [ -n "$db" ] && { echo "# $dt $c"; python $R/tools/rocpd_summary.py $db | grep -E "pw_gemm|pw_head|unit_bf16|stem_pool|dwconv|counter"; } >> $O/r05_shufflenet_pmc_fetch_write.txt
    rm -rf $O/r05_sn_pmc
  done
done
rocprofv3 --kernel-trace --stats -d $O/r05_sn_trace -o t -- python $R/tools/bench_shufflenet.py 128 5 bf16 > /dev/null 2>&1
db=$(find $O/r05_sn_trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/r05_shufflenet_bf16_kernel_trace_stats.txt 2>&1
rm -rf $O/r05_sn_trace
python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --decode-overlap 0 > $O/r05_bench_one_stream.json 2>/dev/null
python $R/tools/latency_b1.py > $O/r05_latency_b1.txt 2>&1
python $R/tools/bench_config5.py > $O/r05_config5.json 2>/dev/null
python $R/tools/bench_tta.py 32 3 > $O/r05_tta.txt 2>&1
python $R/tools/bench_streaming.py > $O/r05_streaming.txt 2>&1
python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype bf16 > $O/r05_bench_bf16.json 2>/dev/null
python $R/tools/profile_layers.py 32 368 368 5 bf16 2>&1 | grep -v amdgpu.ids > $O/r05_bf16_layers.txt
cd $R && python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke.txt 2>&1
# the torch-free C++ host on bench.py's decoder input (scene + 1e-3 * maps): the scene of synth.make_batch as a file
python - <<PY
import importlib, sys
import numpy as np
sys.path.insert(0, "$R")
synth = importlib.import_module("pytorch_realtime_multi-person_pose_estimation_amd.synth")
for n in (32, 11, 8):
    heat, paf, _ = synth.make_batch(n, 368, 368, seed=100)
    with open("/tmp/scene%d.bin" % n, "wb") as f:
        f.write(np.ascontiguousarray(heat, np.float32).tobytes()); f.write(np.ascontiguousarray(paf, np.float32).tobytes())
PY
for args in "32 0 default /tmp/scene32.bin" "11 0 default /tmp/scene11.bin" "8 0 direct /tmp/scene8.bin" "32 2 default /tmp/scene32.bin" "32 1 default /tmp/scene32.bin"; do LD_LIBRARY_PATH=$R/pytorch_realtime_multi-person_pose_estimation_amd/lib $R/examples/c_host $args; done > $O/r05_c_host.txt 2>&1

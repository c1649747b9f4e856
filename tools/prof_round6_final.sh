# round-6 evidence: the bench line (with roofline.traffic and cpu_baseline), rocprofv3 kernel trace of the same command,
# per-launch events, PMC passes (SQ set; FETCH_SIZE; WRITE_SIZE + MFMA counts: separate passes, --kernel-trace only),
# ShuffleNetV2 (configs[3]) per-launch events + trace + FETCH / WRITE passes, and the secondary tools.
# Summaries -> gpurun_out/r06_*; copied into profiles/ by hand.  (The GPU suite: python -m pytest tests -m gpu -q > gpurun_out/r06_gpu_tests.txt.)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
python $R/bench.py --steps 20 --warmup 3 > $O/r06_bench.json 2> $O/r06_bench.err
# the bf16 lines early: session h measured them last, on a box whose bf16 rates had fallen by a third by then
python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype bf16 > $O/r06_bench_bf16.json 2>/dev/null
python $R/tools/bench_tta.py 32 3 > $O/r06_tta.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/r06_trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r06_bench_under_rocprof.json 2> $O/r06_trace.err
db=$(find $O/r06_trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/r06_bench_kernel_trace_stats.txt 2>&1
rm -rf $O/r06_trace
# the decoder's kernels with nothing beside them (one stream), fp32 and bf16 plan: r06_decoder_kernels.txt
: > $O/r06_decoder_kernels.txt
for dt in fp32 bf16; do
  rocprofv3 --kernel-trace --stats -d $O/r06_trace1 -o t -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic --decode-overlap 0 --dtype $dt > /dev/null 2> $O/r06_trace1.err
  db=$(find $O/r06_trace1 -name "*.db" | head -1)
  [ -n "$db" ] && { echo "# bench.py --decode-overlap 0 --dtype $dt: calls, total us, average us, share of the kernel time"; python $R/tools/rocpd_summary.py $db | grep -E "nms_refine|limb_assign|group_kernel|peak_prefix|clear_header|layout_axpby|conv_first|tail_"; } >> $O/r06_decoder_kernels.txt
  rm -rf $O/r06_trace1
done
python $R/tools/profile_layers.py 32 368 368 5 fp32 > $O/r06_fp32_layers.txt 2>&1
: > $O/r06_pmc_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA"; do
  rocprofv3 --pmc $set --kernel-trace -d $O/r06_pmc -o t -- python $R/tools/profile_layers.py 32 368 368 1 fp32 > /dev/null 2>&1
  db=$(find $O/r06_pmc -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db | grep -E "wino|conv_mfma_f32|conv_first|tail_kernel|counter" >> $O/r06_pmc_counters.txt
  rm -rf $O/r06_pmc
done
# ShuffleNetV2 x1.0, 128 x 368 x 368 (configs[3])
for dt in fp32 bf16; do
  VERBOSE=1 python $R/tools/bench_shufflenet.py 128 10 $dt 2>&1 | grep -v amdgpu.ids > $O/r06_shufflenet_${dt}_events.txt
done
rocprofv3 --kernel-trace --stats -d $O/r06_sn_trace -o t -- python $R/tools/bench_shufflenet.py 128 5 fp32 > /dev/null 2>&1
db=$(find $O/r06_sn_trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/r06_shufflenet_kernel_trace_stats.txt 2>&1
rm -rf $O/r06_sn_trace
: > $O/r06_shufflenet_pmc_fetch_write.txt
for dt in fp32 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $O/r06_sn_pmc -o t -- python $R/tools/bench_shufflenet.py 128 2 $dt > /dev/null 2>&1
    db=$(find $O/r06_sn_pmc -name "*.db" | head -1)
    [ -n "$db" ] && { echo "# $dt $c"; python $R/tools/rocpd_summary.py $db | grep -E "pw_gemm|pw_head|unit_bf16|stem_pool|dwconv|counter"; } >> $O/r06_shufflenet_pmc_fetch_write.txt
    rm -rf $O/r06_sn_pmc
  done
done
rocprofv3 --kernel-trace --stats -d $O/r06_sn_trace -o t -- python $R/tools/bench_shufflenet.py 128 5 bf16 > /dev/null 2>&1
db=$(find $O/r06_sn_trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/r06_shufflenet_bf16_kernel_trace_stats.txt 2>&1
rm -rf $O/r06_sn_trace
python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --decode-overlap 0 > $O/r06_bench_one_stream.json 2>/dev/null
python $R/tools/latency_b1.py > $O/r06_latency_b1.txt 2>&1
python $R/tools/bench_config5.py > $O/r06_config5.json 2>/dev/null
python $R/tools/bench_streaming.py > $O/r06_streaming.txt 2>&1
python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-traffic --dtype bf16 > $O/r06_bench_bf16_late.json 2>/dev/null
python $R/tools/profile_layers.py 32 368 368 5 bf16 2>&1 | grep -v amdgpu.ids > $O/r06_bf16_layers.txt
cd $R && python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke.txt 2>&1
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
for args in "32 0 default /tmp/scene32.bin" "11 0 default /tmp/scene11.bin" "8 0 direct /tmp/scene8.bin" "32 2 default /tmp/scene32.bin" "32 1 default /tmp/scene32.bin"; do LD_LIBRARY_PATH=$R/pytorch_realtime_multi-person_pose_estimation_amd/lib $R/examples/c_host $args; done > $O/r06_c_host.txt 2>&1

# the production library's decoder beside the next forward, default guard, 12 decodes per step
cd $R
for dt in "50000 bf16" "5000 fp32 bf16x3"; do REPEATS=12 PEOPLE=8 timeout 900 python tools/exp/overlap_soak.py $dt 2>&1 | grep -E "differing records|serial path"; done > $O/r06_decoder_soak_final_tree.txt

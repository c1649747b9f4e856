# round-3 evidence (fp32 plan: conv1_1 kernel, F(4x4,3x3) for the 3x3 convs, F(6,7) in the 8-wave form):
# GPU tests, the bench line, rocprofv3 kernel trace of the same command, per-launch events, PMC passes (SQ set;
# FETCH_SIZE; WRITE_SIZE + MFMA counts: separate passes, --kernel-trace only), and the secondary tools.
# Summaries -> gpurun_out/r03_*; copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R && python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -4 > $O/r03_gpu_tests.txt; cd /tmp
python $R/bench.py --steps 20 --warmup 3 > $O/r03_bench.json 2> $O/r03_bench.err
rocprofv3 --kernel-trace --stats -d $O/r03_trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r03_bench_under_rocprof.json 2> $O/r03_trace.err
db=$(find $O/r03_trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/r03_bench_kernel_trace_stats.txt 2>&1
rm -rf $O/r03_trace
python $R/tools/profile_layers.py 32 368 368 5 fp32 > $O/r03_fp32_layers.txt 2>&1
: > $O/r03_pmc_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA"; do
  rocprofv3 --pmc $set --kernel-trace -d $O/r03_pmc -o t -- python $R/tools/profile_layers.py 32 368 368 1 fp32 > /dev/null 2>&1
  db=$(find $O/r03_pmc -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db | grep -E "wino|conv_mfma_f32|conv_first|counter" >> $O/r03_pmc_counters.txt
  rm -rf $O/r03_pmc
done
RTPOSE_WINOGRAD3_M=2 python $R/tools/profile_layers.py 32 368 368 5 fp32 > $O/r03_fp32_layers_f23.txt 2>&1
python $R/tools/bench_conv3.py > $O/r03_conv3_forms.txt 2>&1
python $R/tools/latency_b1.py > $O/r03_latency_b1.txt 2>&1
python $R/tools/bench_config5.py > $O/r03_config5.json 2>/dev/null
python $R/tools/bench_tta.py 32 3 > $O/r03_tta.txt 2>&1
python $R/tools/bench_streaming.py > $O/r03_streaming.txt 2>&1
cd $R && python -c "import __graft_entry__ as g; g.smoke()" > $O/r03_smoke.txt 2>&1
for args in "32 0" "11 0" "8 0 auto" "32 2" "32 1"; do LD_LIBRARY_PATH=$R/pytorch_realtime_multi-person_pose_estimation_amd/lib $R/examples/c_host $args; done > $O/r03_c_host.txt 2>&1

# round-2 final evidence (fp32 plan with the Winograd kernels): GPU tests, the bench line, rocprofv3 kernel trace of
# the same command, per-launch events, PMC passes (SQ set; FETCH_SIZE; WRITE_SIZE + MFMA counts: separate passes,
# --kernel-trace only), and the secondary tools.  Summaries -> gpurun_out/r02f_*; copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R && python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/r02f_gpu_tests.txt; cd /tmp
python $R/bench.py --steps 20 --warmup 3 > $O/r02f_bench.json 2> $O/r02f_bench.err
rocprofv3 --kernel-trace --stats -d $O/r02f_trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r02f_bench_under_rocprof.json 2> $O/r02f_trace.err
db=$(find $O/r02f_trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/r02f_bench_kernel_trace_stats.txt 2>&1
rm -rf $O/r02f_trace
python $R/tools/profile_layers.py 32 368 368 5 fp32 > $O/r02f_fp32_layers.txt 2>&1
: > $O/r02f_pmc_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA"; do
  rocprofv3 --pmc $set --kernel-trace -d $O/r02f_pmc -o t -- python $R/tools/profile_layers.py 32 368 368 1 fp32 > /dev/null 2>&1
  db=$(find $O/r02f_pmc -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db | grep -E "wino|conv_mfma_f32|counter" >> $O/r02f_pmc_counters.txt
  rm -rf $O/r02f_pmc
done
python $R/tools/bench_config5.py > $O/r02f_config5.json 2>/dev/null
python $R/tools/bench_streaming.py > $O/r02f_streaming.txt 2>&1
python $R/tools/bench_tta.py 32 3 > $O/r02f_tta.txt 2>&1
python $R/tools/latency_b1.py > $O/r02f_latency_b1.txt 2>&1

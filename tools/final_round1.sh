# round-1 final evidence run: full GPU suite, smoke, fp32 bench (+rocprof trace), bf16 bench, config 5, TTA
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $O/final_gpu_tests.log; cat $O/final_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 3 > $O/final_bench.json 2> $O/final_bench.err; cat $O/final_bench.json
timeout 600 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/final_bench_bf16.json 2> $O/final_bench_bf16.err; cat $O/final_bench_bf16.json
timeout 600 python tools/bench_config5.py > $O/final_config5.json 2> $O/final_config5.err; cat $O/final_config5.json
timeout 600 python tools/bench_config5.py --dtype bf16 >> $O/final_config5.json 2>> $O/final_config5.err; tail -1 $O/final_config5.json
timeout 600 python bench.py --dtype bf16x3 --steps 10 --warmup 3 --no-cpu-baseline > $O/final_bench_bf16x3.json 2> $O/final_bench_bf16x3.err; cut -c1-160 $O/final_bench_bf16x3.json
timeout 600 python tools/bench_config5.py --dtype bf16x3 >> $O/final_config5.json 2>> $O/final_config5.err
timeout 300 python tools/bench_tta.py 32 3 > $O/final_tta.log 2>&1; tail -8 $O/final_tta.log
(timeout 300 python tools/bench_shufflenet.py 128 5 fp32; timeout 300 python tools/bench_shufflenet.py 128 5 bf16) > $O/final_shufflenet.log 2>&1; grep "^launches" $O/final_shufflenet.log
timeout 300 python tools/latency_b1.py > $O/final_latency.log 2>&1; tail -3 $O/final_latency.log
timeout 300 python tools/bench_streaming.py > $O/final_streaming.log 2>&1; tail -3 $O/final_streaming.log
timeout 300 python tools/profile_layers.py 32 368 368 3 bf16x3 > $O/final_bf16x3_layers.log 2>&1
timeout 300 python tools/profile_layers.py 32 368 368 3 bf16 > $O/final_bf16_layers.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/final_trace -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/final_trace_bench.json 2> $O/final_trace.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/final_trace_bf16 -o r1 -- python $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/final_trace_bf16_bench.json 2> $O/final_trace_bf16.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/final_pmc_bf16 -o r1 -- python $R/tools/profile_layers.py 32 368 368 1 bf16 > $O/final_pmc_bf16.log 2>&1
cd $R
for d in final_trace final_trace_bf16 final_pmc_bf16; do
  db=$(find $O/$d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/$d.summary.txt 2>&1
done
find $O -name "*.db" -delete
head -12 $O/final_trace.summary.txt

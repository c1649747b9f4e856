# bf16 path: full GPU suite, TTA timing, bench --dtype bf16, rocprofv3 kernel trace + MFMA PMC
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/bf16_full_gpu_tests.log
cat $O/bf16_full_gpu_tests.log
timeout 300 python tools/bench_tta.py 5 > $O/bf16_tta.log 2>&1; tail -5 $O/bf16_tta.log
timeout 600 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bf16_bench.json 2> $O/bf16_bench.err; cat $O/bf16_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bf16_trace -o r1 -- python $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/prof_bf16_trace_bench.json 2> $O/prof_bf16_trace.err
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/prof_bf16_pmc1 -o r1 -- python $R/tools/profile_layers.py 32 368 368 1 bf16 > $O/prof_bf16_pmc1.log 2>&1
cd $R
for d in prof_bf16_trace prof_bf16_pmc1; do
  db=$(find $O/$d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $O/$d.summary.txt 2>&1
done
find $O -name "*.db" -size +20M -delete
head -30 $O/prof_bf16_trace.summary.txt

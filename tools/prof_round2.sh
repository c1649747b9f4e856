# round-2 evidence: rocprofv3 --kernel-trace --stats over the bench command (the contract line), summaries -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/r02_trace -o t -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r02_bench_under_rocprof.json 2> $O/r02_trace.err
db=$(find $O/r02_trace -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $O/r02_bench_kernel_trace_stats.txt 2>&1
rm -rf $O/r02_trace

cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
: > $O/r05_shufflenet_pmc_fetch_write.txt
for dt in bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $O/r05_sn_pmc -o t -- python $R/tools/bench_shufflenet.py 128 2 $dt > /dev/null 2>&1
    db=$(find $O/r05_sn_pmc -name "*.db" | head -1)
    [ -n "$db" ] && { echo "# $dt $c"; python $R/tools/rocpd_summary.py $db | grep -E "pw_gemm|pw_head|unit_bf16|stem_pool|dwconv|counter"; } >> $O/r05_shufflenet_pmc_fetch_write.txt
    rm -rf $O/r05_sn_pmc
  done
done
cat $O/r05_shufflenet_pmc_fetch_write.txt

#!/bin/bash
# Round-4 session 33: SQ counters of the final ShuffleNetV2 fp32 kernels and of the bf16 conv kernels (transposed-product epilogue)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
SET="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
{
echo "# Round 4, final build: SQ counters of the ShuffleNetV2 fp32 kernels (rocprofv3 --pmc $SET --kernel-trace, tools/bench_shufflenet.py 128 2 fp32; 3 forwards)"
rocprofv3 --pmc $SET --kernel-trace -d $O/s33_pmc -o t -- python $R/tools/bench_shufflenet.py 128 2 fp32 > /dev/null 2>&1
db=$(find $O/s33_pmc -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db | grep -E "pw_gemm|pw_head|stem_pool|dwconv|counter|^kernel"
rm -rf $O/s33_pmc
} > $O/r04_shufflenet_pmc_sq.txt 2>&1
{
echo "# Round 4, final build: SQ counters of the bf16 conv kernels (rocprofv3 --pmc $SET --kernel-trace, tools/profile_layers.py 32 368 368 1 bf16; 2 forwards)"
rocprofv3 --pmc $SET --kernel-trace -d $O/s33_pmc -o t -- python $R/tools/profile_layers.py 32 368 368 1 bf16 > /dev/null 2>&1
db=$(find $O/s33_pmc -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_summary.py $db | grep -E "conv_mfma_bf16<[37]|counter|^kernel"
rm -rf $O/s33_pmc
} > $O/r04_bf16_pmc_sq.txt 2>&1
cat $O/r04_shufflenet_pmc_sq.txt | cut -c1-160 | head -70; cat $O/r04_bf16_pmc_sq.txt | cut -c1-160 | grep "7, 32, 0" 

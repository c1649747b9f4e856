cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof_trace -o r1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/prof_trace_bench.json 2> $O/prof_trace.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/prof_pmc1 -o r1 -- python $R/tools/profile_layers.py 32 368 368 1 > $O/prof_pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace -d $O/prof_pmc2 -o r1 -- python $R/tools/profile_layers.py 32 368 368 1 > $O/prof_pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA --kernel-trace -d $O/prof_pmc3 -o r1 -- python $R/tools/profile_layers.py 32 368 368 1 > $O/prof_pmc3.log 2>&1
ls -R $O | head -50

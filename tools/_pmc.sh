# PMC passes for the dominant kernel alone (tools/prof_wino.py runs conv3x3_wino_kernel 4 x on the config-2 first-layer
# shape).  One counter group per run; no TCP_*_LATENCY counters (they hang the profiler on this pool).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcw
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/p1 --output-format csv -- python3 $R/tools/prof_wino.py $1 $2 $3 $4 > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/p2 --output-format csv -- python3 $R/tools/prof_wino.py $1 $2 $3 $4 > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU -d $O/p3 --output-format csv -- python3 $R/tools/prof_wino.py $1 $2 $3 $4 > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/p4 --output-format csv -- python3 $R/tools/prof_wino.py $1 $2 $3 $4 > $O/p4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/p5 --output-format csv -- python3 $R/tools/prof_wino.py $1 $2 $3 $4 > $O/p5.log 2>&1
ls $O

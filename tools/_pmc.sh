set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcw
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/p1 --output-format csv -- python3 $R/tools/prof_wino.py > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/p2 --output-format csv -- python3 $R/tools/prof_wino.py > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCP_LATENCY TCP_PENDING_STALL_CYCLES TA_BUSY TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY TA_TOTAL_WAVEFRONTS -d $O/p3 --output-format csv -- python3 $R/tools/prof_wino.py > $O/p3.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL TCC_BUSY TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL -d $O/p4 --output-format csv -- python3 $R/tools/prof_wino.py > $O/p4.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_LDS SQ_INST_LEVEL_LDS -d $O/p5 --output-format csv -- python3 $R/tools/prof_wino.py > $O/p5.log 2>&1
ls $O/*

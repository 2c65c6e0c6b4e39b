# per-map-size PMC table of the Winograd kernels -> gpurun_out/pmc_wino_table/{sq,fetch,write} + r03_wino_pmc_table.json
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_wino_table
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $O/sq --output-format csv -- python3 $R/tools/wino_pmc_table.py run > $O/sq.log 2>&1
echo sq done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/tools/wino_pmc_table.py run > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/tools/wino_pmc_table.py run > $O/write.log 2>&1
echo write done
python3 $R/tools/wino_pmc_table.py parse $O > $R/gpurun_out/r03_wino_pmc_table.json
cat $R/gpurun_out/r03_wino_pmc_table.json

# PMC passes over ONE eager registration iteration of the default bench workload (every kernel of the path in its real
# shapes): FETCH_SIZE, WRITE_SIZE (separate passes, MI355X_MICROARCH.md) and an SQ/GRBM pass for the matrix-pipe duty.
# Side streams off (CMR_STREAMS=0) so that a dispatch's counters are its own.  -> gpurun_out/pmc_path/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_path
rm -rf $O; mkdir -p $O
export CMR_STREAMS=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-train-lines --no-pipeline-line > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-train-lines --no-pipeline-line > $O/write.log 2>&1
echo write done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/sq --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-train-lines --no-pipeline-line > $O/sq.log 2>&1
echo sq done
python3 $R/tools/pmc_path.py $O > $O/pmc_path.json
head -c 1500 $O/pmc_path.json

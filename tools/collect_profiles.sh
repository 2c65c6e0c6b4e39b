# Runs on the GPU box (gpurun -- 'bash tools/collect_profiles.sh'): kernel-trace statistics of the default bench run and
# the two HBM-traffic PMC passes (separate runs, as MI355X_MICROARCH.md prescribes).  Summaries land in gpurun_out/prof_r/;
# tools/pmc_traffic.py turns the counter CSVs into profiles/*_pmc_traffic.json.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline > $O/write.log 2>&1
echo write done
ls $O/*/*

# Round-3 profiles (GPU box): rocprofv3 kernel statistics of (a) the registration part of the default bench command alone
# (--no-train-lines --no-pipeline-line --no-cpu-baseline: the launches the JSON line's roofline averages over) and (b) the exact default
# command (python bench.py: also the pipelined re-measurement and the train / train_geo sub-objects), then the PMC passes of tools/_pmc_path.sh.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/reg --output-format csv -- python3 $R/bench.py --no-train-lines --no-pipeline-line --no-cpu-baseline > $O/bench_register_only_under_rocprof.json 2> $O/reg.err
echo reg done
rocprofv3 --kernel-trace --stats -d $O/full --output-format csv -- python3 $R/bench.py > $O/bench_default_under_rocprof.json 2> $O/full.err
echo full done
cp $O/reg/*/*kernel_stats.csv $O/kernel_stats_register_only.csv
cp $O/full/*/*kernel_stats.csv $O/kernel_stats_default_command.csv
rm -rf $O/reg $O/full
bash $R/tools/_pmc_path.sh > $O/pmc_path.log 2>&1
cp $R/gpurun_out/pmc_path/pmc_path.json $O/pmc_path.json
ls -la $O

#!/bin/bash
# kernel-trace statistics of the agent-update bench (bench.py --mode train --dtype bf16) -> gpurun_out/prof_r05/kernel_stats_train.csv
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r05; mkdir -p $O; rm -rf $O/tr
( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/tr --output-format csv -- python3 $R/bench.py --mode train --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_under_rocprof.json 2> $O/tr.err ) && echo train done
cp $O/tr/*/*kernel_stats.csv $O/kernel_stats_train.csv && rm -rf $O/tr
head -32 $O/kernel_stats_train.csv | cut -c1-150

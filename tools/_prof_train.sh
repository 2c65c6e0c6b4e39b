# kernel-trace statistics of the agent-update bench (bench.py --mode train) -> gpurun_out/prof_train/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_train
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 $R/bench.py --mode train --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -30 $O/kernel_stats.csv | cut -c1-200

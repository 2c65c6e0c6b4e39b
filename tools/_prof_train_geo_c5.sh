# kernel-trace statistics of the geometric-model update at SURVEY.md 8d's C5 shape (352x1216, 65 536 points, point prologue in the step)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_train_geo_c5
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 $R/bench.py --mode train-geo --num-pt 65536 --img 352x1216 --prologue --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
head -5 $O/kernel_stats.csv | cut -c1-200

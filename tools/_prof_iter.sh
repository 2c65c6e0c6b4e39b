set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_iter
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 $R/bench.py --mode iter --steps 5 --warmup 2 > $O/bench.json 2> $O/err.txt
cp $O/*/*kernel_stats.csv $O/kernel_stats.csv
head -12 $O/kernel_stats.csv | cut -c1-150

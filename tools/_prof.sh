set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_c
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
ls $O/*/

# kernel trace of `bench.py --replay-only` (timed mode only) for one workload -> gpurun_out/r06_replay_census_<tag>.txt (tools/replay_census.py)
# bash tools/_trace_replay.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/trace_replay_$tag
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O --output-format csv -- python3 $R/bench.py --replay-only --steps 4 --warmup 2 --no-cpu-baseline "$@" > $O/bench.json 2> $O/bench.err
python3 $R/tools/replay_census.py $O > $R/gpurun_out/r06_replay_census_$tag.txt
head -30 $R/gpurun_out/r06_replay_census_$tag.txt
python3 $R/tools/train_timeline.py $O 1 stem_a_kernel > $R/gpurun_out/r06_replay_timeline_$tag.txt
rm -rf $O/*/

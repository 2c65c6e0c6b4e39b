# kernel trace of the default bench (hipGraph replay) -> gpurun_out/trace_graph/ ; tools/trace_gaps.py summarises the last iteration
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_graph
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
ls $O/*/ | head

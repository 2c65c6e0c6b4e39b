#!/bin/bash
# kernel trace of the bf16 agent update (hipGraph replay) and its timeline -> gpurun_out/r06_train_timeline.txt
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_train
rm -rf $O; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace -d $O --output-format csv -- python3 $R/bench.py --mode train --dtype bf16 --steps 6 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err )
python3 tools/train_timeline.py $O 2 > gpurun_out/r06_train_timeline.txt
head -40 gpurun_out/r06_train_timeline.txt
rm -rf $O

#!/bin/bash
# tools/side_queue_probe4.py under rocprofv3 --kernel-trace -> gpurun_out/r06_side_queue_probe4.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_probe4
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O --output-format csv -- python3 $R/tools/side_queue_probe4.py > $O/out.txt 2> $O/err.txt
python3 $R/tools/side_queue_probe4_parse.py $O > $R/gpurun_out/r06_side_queue_probe4.txt
rm -rf $O/*/
cat $R/gpurun_out/r06_side_queue_probe4.txt

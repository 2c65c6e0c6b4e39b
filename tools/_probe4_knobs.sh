#!/bin/bash
# tools/side_queue_probe4.py (80 origin nodes, main first) under the runtime's queue / graph knobs -> gpurun_out/r06_side_queue_probe4_knobs.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PROBE4=80:main_first
out=$R/gpurun_out/r06_side_queue_probe4_knobs.txt
: > $out
for kv in X=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 DEBUG_CLR_MAX_BATCH_SIZE=1 DEBUG_HIP_FORCE_GRAPH_QUEUES=4 AMD_DIRECT_DISPATCH=0 ROC_CPU_WAIT_FOR_SIGNAL=1 DEBUG_CLR_BATCH_CPU_SYNC_SIZE=1 GPU_STREAMOPS_CP_WAIT=0 DEBUG_HIP_DYNAMIC_QUEUES=0 HIP_FORCE_DEV_KERNARG=0; do
  O=$R/gpurun_out/trace_probe4; rm -rf $O; mkdir -p $O
  export $kv
  rocprofv3 --kernel-trace -d $O --output-format csv -- python3 $R/tools/side_queue_probe4.py > $O/out.txt 2> $O/err.txt
  echo "== $kv" >> $out
  python3 $R/tools/side_queue_probe4_parse.py $O | tail -3 >> $out
  unset ${kv%%=*}
  rm -rf $O
done
cat $out

#!/bin/bash
# round 6: does the HIP runtime's graph executor decide how the two towers of the agent update overlap?  bench.py --mode train --dtype bf16 under
# the CLR graph knobs, same box -> gpurun_out/r06_graph_knobs.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_graph_knobs.txt
: > $out
run() {
  ms=$(env "$@" timeout -k 10 200 python bench.py --mode train --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline $EXTRA 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
  echo "$* -> ms_per_step $ms" | tee -a $out
}
run X=0
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=16
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_MAX_BATCH_SIZE=1
run DEBUG_HIP_FORCE_ASYNC_QUEUE=1
run X=0
EXTRA=--eager run X=0

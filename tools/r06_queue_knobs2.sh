#!/bin/bash
# round 6: hardware-queue counts under ROC_CPU_WAIT_FOR_SIGNAL=1 (bench.py's default now): replayed headline and c3 -> gpurun_out/r06_queue_knobs2.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_queue_knobs2.txt
: > $out
run() {
  kv=$1; shift
  ms=$(env $kv timeout -k 10 200 python bench.py --replay-only --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f' % d['ms_per_step'])")
  echo "$kv $* -> ms_per_step $ms" | tee -a $out
}
for kv in X=0 GPU_MAX_HW_QUEUES=3 GPU_MAX_HW_QUEUES=5 GPU_MAX_HW_QUEUES=6 DEBUG_HIP_FORCE_GRAPH_QUEUES=2 DEBUG_HIP_FORCE_GRAPH_QUEUES=3 DEBUG_HIP_FORCE_GRAPH_QUEUES=6 ROC_ACTIVE_WAIT_TIMEOUT=100 ROC_ACTIVE_WAIT_TIMEOUT=10000 X=0; do
  run $kv
  run $kv --workload c3
done

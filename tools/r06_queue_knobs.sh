#!/bin/bash
# round 6: the side queue of a replayed registration graph starts 0.6 - 0.95 ms after the graph and its first ~16 kernels come every ~62 us
# (profiles/r06_replay_timeline_*.txt).  Is that the runtime's queue / signal handling?  bench.py --replay-only under the CLR / ROC queue knobs,
# same box -> gpurun_out/r06_queue_knobs.txt      usage: bash tools/r06_queue_knobs.sh [bench args, e.g. --workload c3]
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_queue_knobs.txt
echo "# bench.py --replay-only $*" >> $out
run() {
  ms=$(env "$@" timeout -k 10 200 python bench.py --replay-only --steps 20 --warmup 5 --no-cpu-baseline $ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f' % d['ms_per_step'])")
  echo "$* -> ms_per_step $ms" | tee -a $out
}
ARGS="$*"
run X=0
run ROC_ACTIVE_WAIT_TIMEOUT=2000
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run DEBUG_HIP_DYNAMIC_QUEUES=0
run DEBUG_HIP_DYNAMIC_QUEUES=1
run GPU_STREAMOPS_CP_WAIT=0
run GPU_STREAMOPS_CP_WAIT=1
run ROC_CPU_WAIT_FOR_SIGNAL=0
run ROC_CPU_WAIT_FOR_SIGNAL=1
run ROC_SYSTEM_SCOPE_SIGNAL=0
run AMD_DIRECT_DISPATCH=0
run CMR_STREAMS_MAIN_FIRST=none
run HIP_FORCE_DEV_KERNARG=1
run X=0

#!/bin/bash
# round 6: ROC_CPU_WAIT_FOR_SIGNAL=1 (the runtime resolves cross-queue waits on the host instead of with barrier packets) moved the c3 replay
# by - 1.7 % in the knob sweep (profiles/r06_queue_knobs.txt); all lines, same box, alternating -> gpurun_out/r06_ab_cpuwait.txt
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_ab_cpuwait.txt
: > $O
run() {  # run <label> <env> <bench args>
  lab=$1; kv=$2; shift 2
  ms=$(env $kv timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f' % d['ms_per_step'])")
  echo "$lab $kv -> ms_per_step $ms" | tee -a $O
}
for rep in 1 2; do
  for v in 0 1; do
    run "c3       " ROC_CPU_WAIT_FOR_SIGNAL=$v --replay-only --workload c3
    run "c1 bf16  " ROC_CPU_WAIT_FOR_SIGNAL=$v --replay-only --dtype bf16
    run "c1 f32   " ROC_CPU_WAIT_FOR_SIGNAL=$v --replay-only
    run "train    " ROC_CPU_WAIT_FOR_SIGNAL=$v --mode train --dtype bf16
    run "train-geo" ROC_CPU_WAIT_FOR_SIGNAL=$v --mode train-geo --num-pt 65536 --steps 5 --warmup 2
  done
done

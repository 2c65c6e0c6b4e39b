#!/bin/bash
# round 6: cross-stream edges of a replayed graph appear to resolve at the granularity of the origin stream's command batches
# (profiles/r06_train_timeline*.txt): the agent update under DEBUG_CLR_MAX_BATCH_SIZE / DEBUG_CLR_BATCH_CPU_SYNC_SIZE -> gpurun_out/r06_batch_knob.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_batch_knob.txt
: > $out
run() {
  ms=$(env "$@" timeout -k 10 200 python bench.py --mode train --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f' % d['ms_per_step'])")
  echo "$* -> ms_per_step $ms" | tee -a $out
}
run X=0
for b in 2 4 8 16 32 128 512; do run DEBUG_CLR_MAX_BATCH_SIZE=$b; done
for b in 4 16 64; do run DEBUG_CLR_BATCH_CPU_SYNC_SIZE=$b; done
run ROC_SIGNAL_POOL_SIZE=4096
run X=0

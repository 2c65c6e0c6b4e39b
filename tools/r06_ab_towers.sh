#!/bin/bash
# round 6: towers fork with the image tower as the side branch issued first (CMR_TOWERS_SWAPPED=1) against the round-3 arrangement (0):
# bench.py --replay-only for the three registration lines, same box, alternating -> gpurun_out/r06_ab_towers.txt
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_ab_towers.txt
: > $O
run() {  # run <label> <env> <bench args>
  lab=$1; kv=$2; shift 2
  ms=$(env $kv timeout -k 10 200 python bench.py --replay-only --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>>gpurun_out/r06_ab_towers.err | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f' % d['ms_per_step'])")
  echo "$lab $kv -> ms_per_step $ms" | tee -a $O
}
for rep in 1 2; do
  for v in 0 1; do
    run "c3      " CMR_TOWERS_SWAPPED=$v --workload c3
    run "c1 bf16 " CMR_TOWERS_SWAPPED=$v --dtype bf16
    run "c1 f32  " CMR_TOWERS_SWAPPED=$v
  done
done

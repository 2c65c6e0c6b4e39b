#!/bin/bash
# round 6: which forks of the registration graph still pay under ROC_CPU_WAIT_FOR_SIGNAL=1?  bench.py --replay-only (fp32 headline and c3)
# with subsets of the tagged forks (CMR_STREAMS_ONLY) -> gpurun_out/r06_ab_forks.txt
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_ab_forks.txt
: > $O
run() {  # run <label> <env> <bench args>
  lab=$1; kv=$2; shift 2
  ms=$(env $kv timeout -k 10 200 python bench.py --replay-only --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f' % d['ms_per_step'])")
  echo "$lab $kv -> ms_per_step $ms" | tee -a $O
}
for rep in 1 2; do
  for only in "" towers,agent towers,agent,coarse_sa towers,agent,fine_sa,fuse towers,agent,heads,trunk towers,agent,coarse_sa,fine_sa,fuse,heads towers,coarse_sa,fine_sa,fuse,heads,trunk; do
    run "c1 f32" CMR_STREAMS_ONLY=$only
    run "c3    " CMR_STREAMS_ONLY=$only --workload c3
  done
done

#!/bin/bash
# round 6: the bf16 agent update with its two towers captured branch after branch (CMR_STREAMS_INTERLEAVE=0) / interleaved (1), same box, alternating
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_ab_interleave.txt
: > $out
for rep in 1 2 3; do
for il in 0 1; do
  for dt in bf16 f32; do
  ms=$(CMR_STREAMS_INTERLEAVE=$il timeout -k 10 200 python bench.py --mode train --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
  echo "rep $rep  interleave=$il dtype=$dt  ms_per_step $ms" | tee -a $out
  done
done
done

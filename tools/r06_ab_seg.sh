#!/bin/bash
# round 6: the agent update replayed as ONE multi-branch hipGraph (CMR_SEGMENTED_GRAPH=0) / as a program of single-chain graphs (1), same box, alternating
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_ab_seg.txt
: > $out
for rep in 1 2 3; do
for sg in 0 1; do
  for dt in bf16 f32; do
  ms=$(CMR_SEGMENTED_GRAPH=$sg timeout -k 10 200 python bench.py --mode train --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r06_ab_seg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
  echo "rep $rep  segmented=$sg dtype=$dt  ms_per_step $ms" | tee -a $out
  done
done
done
tail -5 gpurun_out/r06_ab_seg.err

#!/bin/bash
# round 6: the registration step replayed as ONE multi-branch hipGraph (CMR_SEGMENTED_GRAPH_REG=0) / as a program of single-chain graphs (1)
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_ab_seg_reg.txt
: > $out
for rep in 1 2; do
for sg in 0 1; do
  for dt in f32 bf16; do
  ms=$(CMR_SEGMENTED_GRAPH_REG=$sg timeout -k 10 300 python bench.py --replay-only --no-cpu-baseline --steps 10 --warmup 3 --dtype $dt 2>gpurun_out/r06_ab_seg_reg.err | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.1f it/s' % (d['ms_per_step'], d['value']))")
  echo "rep $rep  segmented=$sg dtype=$dt  $ms" | tee -a $out
  done
done
done
tail -3 gpurun_out/r06_ab_seg_reg.err

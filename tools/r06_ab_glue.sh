#!/bin/bash
# round 6: the glue launch between the blocks of the agent's 3-D branch (cmr_colmax_bias2_f32, CMR_COLMAX_BIAS2=1 default) against the three
# launches it replaces (0): default line (fp32 headline + c3 + c1_bf16), same box, alternating -> gpurun_out/r06_ab_glue.txt
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_ab_glue.txt
: > $O
for rep in 1 2; do
  for v in 1 0; do
    CMR_COLMAX_BIAS2=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>>gpurun_out/r06_ab_glue.err | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('rep $rep  glue=$v  headline %.1f it/s (%.3f ms)  c3 %.1f it/s (%.3f ms)  c1_bf16 %.1f it/s (%.3f ms)' % (d['value'], d['ms_per_step'], d['c3']['value'], d['c3']['ms_per_step'], d['c1_bf16']['value'], d['c1_bf16']['ms_per_step']))" | tee -a $O
  done
done

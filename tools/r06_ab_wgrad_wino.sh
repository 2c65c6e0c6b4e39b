#!/bin/bash
# round 6: the geometric update with the 64 -> 64 weight gradients in the Winograd domain (CMR_WGRAD_WINO=1, default) and on the direct kernel (0),
# same box, alternating: C5 (352x1216, 65 536 points, prologue in the step) and the 160x512 step -> gpurun_out/r06_ab_wgrad_wino.txt
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_ab_wgrad_wino.txt
: > $O
for rep in 1 2; do
  for v in 1 0; do
    ms=$(CMR_WGRAD_WINO=$v timeout -k 10 200 python bench.py --mode train-geo --num-pt 65536 --img 352x1216 --prologue --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $rep  C5 352x1216  wgrad_wino=$v  ms_per_step $ms" | tee -a $O
    ms=$(CMR_WGRAD_WINO=$v timeout -k 10 200 python bench.py --mode train-geo --num-pt 65536 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $rep  160x512      wgrad_wino=$v  ms_per_step $ms" | tee -a $O
  done
done

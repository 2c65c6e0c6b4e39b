#!/bin/bash
# round 6: the bf16 agent update (bench.py --mode train --dtype bf16) with the linear + BatchNorm layers' products in fp32 / bf16, forward and
# backward switched separately, same box, alternating -> gpurun_out/r06_ab_bnl.txt
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_ab_bnl.txt
: > $out
for rep in 1 2; do
for cfg in "0 0" "0 1" "1 1" "1 0"; do
  set -- $cfg
  ms=$(CMR_BN_LINEAR_BF16_FWD=$1 CMR_BN_LINEAR_BF16_BWD=$2 timeout -k 10 200 python bench.py --mode train --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
  echo "rep $rep  fwd_bf16=$1 bwd_bf16=$2  ms_per_step $ms" | tee -a $out
done
done

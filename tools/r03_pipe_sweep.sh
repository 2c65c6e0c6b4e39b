#!/bin/bash
# pipeline CU-budget sweep (geo stage, agent stage)
cd "$GRAFT_REPO_ROOT"
for b in "0,0" "208,48" "192,64" "160,96" "128,128" "224,0" "0,64" "0,128" "224,224"; do
  CMR_PIPE_BUDGET=$b timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-lines --pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('budget $b', round(d['value'],1), round(d['ms_per_step'],3))"
done

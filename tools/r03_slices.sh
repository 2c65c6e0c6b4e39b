#!/bin/bash
# round 3: time-sliced Winograd launches of the image tower (CMR_TOWER_SLICES) x high-priority side streams (CMR_SIDE_PRIORITY), fp32 headline
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/r03_slices.txt
for rep in 1 2; do
for cfg in "1 0" "2 0" "4 0" "8 0" "1 1" "4 1" "8 1" "16 1"; do
  set -- $cfg
  CMR_TOWER_SLICES=$1 CMR_SIDE_PRIORITY=$2 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-lines --no-pipeline-line > gpurun_out/sl.json 2> gpurun_out/sl.err
  python - <<EOF2 | tee -a gpurun_out/r03_slices.txt
import json
d=json.loads(open("gpurun_out/sl.json").read().strip().splitlines()[-1])
print("slices $1 side-priority $2 :", round(d["value"],1), "it/s", round(d["ms_per_step"],3), "ms")
EOF2
done
done

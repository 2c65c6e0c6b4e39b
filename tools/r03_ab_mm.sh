#!/bin/bash
# round 3: A/B of the matrix-class bf16 convolution in the whole step, same box, alternating
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
for i in 1 2; do
  for v in 1 0; do
    CMR_B16_MM=$v timeout -k 10 300 python bench.py --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --no-train-lines > gpurun_out/ab_mm_$v.json 2> gpurun_out/ab_mm_$v.err
    python - <<EOF2
import json
d=json.loads(open("gpurun_out/ab_mm_$v.json").read().strip().splitlines()[-1])
print("c1 bf16 mm=$v", round(d["value"],1), round(d["ms_per_step"],3), "pipelined", round((d.get("pipelined") or {}).get("value",0),1))
EOF2
  done
done
for v in 1 0; do
  CMR_B16_MM=$v timeout -k 10 300 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-train-lines > gpurun_out/ab_mm3_$v.json 2> gpurun_out/ab_mm3_$v.err
  python - <<EOF2
import json
d=json.loads(open("gpurun_out/ab_mm3_$v.json").read().strip().splitlines()[-1])
print("c3 bf16 mm=$v", round(d["value"],1), round(d["ms_per_step"],3), "pipelined", round((d.get("pipelined") or {}).get("value",0),1))
EOF2
done

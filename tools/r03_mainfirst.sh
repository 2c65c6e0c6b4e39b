#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for mf in "towers" "none" "towers,agent,coarse_sa,heads"; do
  for dt in f32 bf16; do
  CMR_STREAMS_MAIN_FIRST=$mf timeout -k 10 200 python bench.py --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --no-train-lines > gpurun_out/mf.json 2> gpurun_out/mf.err
  python - <<EOF2
import json
d=json.loads(open("gpurun_out/mf.json").read().strip().splitlines()[-1])
print("main-first=$mf $dt :", round(d["value"],1), "it/s", round(d["ms_per_step"],3), "ms  pipelined", round((d.get("pipelined") or {}).get("value",0),1))
EOF2
  done
done
done

#!/bin/bash
# round 3: headline + bf16 lines, quick (no CPU baseline, no train lines)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
q() { name=$1; shift; timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-train-lines > gpurun_out/q_$name.json 2> gpurun_out/q_$name.err; python - <<EOF2
import json
d=json.loads(open("gpurun_out/q_$name.json").read().strip().splitlines()[-1])
print("$name", round(d["value"],1), "it/s", round(d["ms_per_step"],3), "ms; pipelined", round((d.get("pipelined") or {}).get("value",0),1), "frac", round(d["roofline"]["frac"],3))
EOF2
}
q c1_f32 --steps 20 --warmup 3
q c1_bf16 --dtype bf16 --steps 20 --warmup 3
q c3_bf16 --workload c3 --steps 10 --warmup 2
q c1_f32_again --steps 20 --warmup 3

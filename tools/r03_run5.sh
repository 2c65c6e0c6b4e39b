#!/bin/bash
# round 3: bf16 lines after the matrix-class convolution kernel
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > gpurun_out/r03_bench_$name.json 2> gpurun_out/r03_bench_$name.err; echo "$name rc=$?"; python - <<EOF2
import json
try:
    d=json.loads(open("gpurun_out/r03_bench_$name.json").read().strip().splitlines()[-1]); r=d.get("roofline",{})
    print("   ", round(d["value"],1), d["unit"], round(d["ms_per_step"],3), "ms; frac", round(r.get("frac",0),3), r.get("bound"), "path", round(r.get("path",0),3), "pipelined", (d.get("pipelined") or {}).get("value"))
except Exception as e: print("   ERR", e)
EOF2
}
run c1_bf16 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline
run c3_bf16 --workload c3 --steps 10 --warmup 2 --no-cpu-baseline
run train_bf16 --mode train --dtype bf16 --steps 20 --warmup 3
timeout -k 10 300 python tools/phases.py sub bf16 > gpurun_out/r03_phases_bf16.txt 2> gpurun_out/r03_phases_bf16.err; cat gpurun_out/r03_phases_bf16.txt

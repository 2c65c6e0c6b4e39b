#!/bin/bash
# round 3: phase timings (fp32 / bf16), launch census of the geo update, default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 python tools/phases.py sub > gpurun_out/r03_phases_f32.txt 2> gpurun_out/r03_phases_f32.err; echo "phases f32 rc=$?"
timeout -k 10 300 python tools/phases.py sub bf16 > gpurun_out/r03_phases_bf16.txt 2> gpurun_out/r03_phases_bf16.err; echo "phases bf16 rc=$?"
timeout -k 10 300 python tools/geo_launches.py 65536 > gpurun_out/r03_geo_calls.txt 2> gpurun_out/r03_geo_calls.err; echo "geo calls rc=$?"
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "bench rc=$?"
cat gpurun_out/r03_phases_f32.txt
tail -c 600 gpurun_out/r03_bench_default.json

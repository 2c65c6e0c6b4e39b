#!/bin/bash
# round 3: final artefacts -- default line, secondary lines, phases, rocprofv3 kernel statistics of the default command and of the registration part
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "default rc=$?"
bash tools/r03_run3.sh
timeout -k 10 300 python tools/phases.py sub > gpurun_out/r03_phases_f32.txt 2> /dev/null
timeout -k 10 300 python tools/phases.py sub bf16 > gpurun_out/r03_phases_bf16.txt 2> /dev/null
cat gpurun_out/r03_phases_f32.txt gpurun_out/r03_phases_bf16.txt

#!/bin/bash
# round 3, full check: GPU tier, default bench line (with train sub-objects + pipelined), profile of the same command
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
echo "== full gpu tier"; timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t_all.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r03_t_all.log
echo "== bench default"; timeout -k 10 600 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "rc=$?"; tail -c 300 gpurun_out/r03_bench_default.json; tail -2 gpurun_out/r03_bench_default.err

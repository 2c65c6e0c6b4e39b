#!/bin/bash
# round 3, first GPU call: new tests, whole GPU tier, launcher rehearsal, default bench line
set -o pipefail
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
echo "== nested capture (torch only)"; HIP_VISIBLE_DEVICES=0 timeout -k 10 120 python tools/nested_capture_min.py > gpurun_out/r03_nested_capture.txt 2>&1; echo "rc=$?" >> gpurun_out/r03_nested_capture.txt; tail -3 gpurun_out/r03_nested_capture.txt
echo "== new tests"; timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_train_gpu.py tests/test_dropout_gpu.py -x -q -m gpu -k "fps or sgd or dropout_mask or adam" > gpurun_out/r03_t_new.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r03_t_new.log
echo "== bench default"; timeout -k 10 600 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err; echo "rc=$?"; tail -c 600 gpurun_out/r03_bench_default.json; tail -3 gpurun_out/r03_bench_default.err
echo "== bench --gpus 2 gloo share-gpu (register)"; timeout -k 10 600 python bench.py --gpus 2 --dist-backend gloo --share-gpu --steps 5 --warmup 2 > gpurun_out/r03_bench_2rank_register.json 2> gpurun_out/r03_bench_2rank_register.err; echo "rc=$?"; tail -c 400 gpurun_out/r03_bench_2rank_register.json; tail -3 gpurun_out/r03_bench_2rank_register.err
echo "== bench --gpus 2 gloo share-gpu (train)"; timeout -k 10 600 python bench.py --gpus 2 --mode train --dist-backend gloo --share-gpu --steps 5 --warmup 2 > gpurun_out/r03_bench_2rank_train.json 2> gpurun_out/r03_bench_2rank_train.err; echo "rc=$?"; tail -c 600 gpurun_out/r03_bench_2rank_train.json; tail -3 gpurun_out/r03_bench_2rank_train.err
echo "== full gpu tier"; timeout -k 10 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t_all.log 2>&1; echo "rc=$?"; tail -5 gpurun_out/r03_t_all.log

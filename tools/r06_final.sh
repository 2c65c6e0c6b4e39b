#!/bin/bash
# round 6: final artefacts, part 1 (GPU box) -- rocprofv3 kernel statistics of (a) the timed mode alone (hipGraph replays: roofline.frac),
# (b) the registration part of the default command, (c) the agent update, (d) the geometric update (160x512 and the C5 shape); phases; the
# PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) behind every roofline.traffic of the driver line.  Everything lands in gpurun_out/;
# tools/r06_collect.sh (part 2, here in the container) copies it to profiles/r06_* and writes the freshness sidecars; part 3 is the
# default line: gpurun -- 'python bench.py > gpurun_out/r06_bench_default.json'.
#   gpurun --timeout 1200 -- 'bash tools/r06_final.sh'
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
SECONDS=0
prof() {   # prof <tag> <bench args...>
  tag=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/$tag --output-format csv -- python3 $R/bench.py "$@" > $O/bench_${tag}_under_rocprof.json 2> $O/$tag.err ) && echo "$tag done at ${SECONDS}s"
  cp $O/$tag/*/*kernel_stats.csv $O/kernel_stats_$tag.csv && rm -rf $O/$tag
}
prof replay_only --replay-only --no-cpu-baseline &&
prof register_only --no-cpu-baseline --no-train-lines &&
prof train --mode train --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline &&
prof train_geo --mode train-geo --num-pt 65536 --steps 5 --warmup 2 --no-cpu-baseline &&
prof train_geo_c5 --mode train-geo --num-pt 65536 --img 352x1216 --prologue --steps 5 --warmup 2 --no-cpu-baseline &&
timeout -k 10 300 python tools/phases.py sub > gpurun_out/r06_phases_f32.txt 2> /dev/null
cat gpurun_out/r06_phases_f32.txt
echo "kernel statistics done at ${SECONDS}s"
timeout -k 10 500 bash tools/_pmc_path.sh > gpurun_out/r06_pmc_path.log 2>&1 && echo "pmc path done at ${SECONDS}s" &&
timeout -k 10 600 bash tools/_pmc_train.sh > gpurun_out/r06_pmc_train.log 2>&1 && echo "pmc train done at ${SECONDS}s" &&
timeout -k 10 500 bash tools/_pmc_path_bf16.sh > gpurun_out/r06_pmc_bf16.log 2>&1 && echo "pmc bf16 done at ${SECONDS}s"
# keep what part 2 needs, drop the raw counter directories (tens of MB)
cp gpurun_out/pmc_path/pmc_path.json gpurun_out/r06_pmc_path.json 2> /dev/null
cp gpurun_out/pmc_train/agent.json gpurun_out/r06_pmc_train.json 2> /dev/null
cp gpurun_out/pmc_train/geo.json gpurun_out/r06_pmc_train_geo.json 2> /dev/null
cp gpurun_out/pmc_train/c5.json gpurun_out/r06_pmc_train_geo_c5.json 2> /dev/null
rm -rf gpurun_out/pmc_path gpurun_out/pmc_train gpurun_out/pmc_bf16_c3 gpurun_out/pmc_bf16_c1
ls -la $O gpurun_out/*.json | tail -30
echo "all done at ${SECONDS}s"

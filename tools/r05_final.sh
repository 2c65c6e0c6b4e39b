#!/bin/bash
# round 5: final artefacts -- default line; rocprofv3 kernel statistics of (a) the timed mode alone (hipGraph replays: roofline.frac_in_graph),
# (b) the registration part of the default command, (c) the agent update, (d) the geometric update (160x512 and the C5 shape); phases.
# Run through gpurun from the repository root:  gpurun --timeout 1200 -- 'bash tools/r05_final.sh'
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r05
rm -rf $O; mkdir -p $O
SECONDS=0
prof() {   # prof <tag> <bench args...>
  tag=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/$tag --output-format csv -- python3 $R/bench.py "$@" > $O/bench_${tag}_under_rocprof.json 2> $O/$tag.err ) && echo "$tag done at ${SECONDS}s"
  cp $O/$tag/*/*kernel_stats.csv $O/kernel_stats_$tag.csv && rm -rf $O/$tag
}
prof replay_only --replay-only --no-cpu-baseline
prof register_only --no-cpu-baseline --no-train-lines
prof train --mode train --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline
prof train_geo --mode train-geo --num-pt 65536 --steps 5 --warmup 2 --no-cpu-baseline
prof train_geo_c5 --mode train-geo --num-pt 65536 --img 352x1216 --prologue --steps 5 --warmup 2 --no-cpu-baseline
timeout -k 10 300 python tools/phases.py sub > gpurun_out/r05_phases_f32.txt 2> /dev/null
cat gpurun_out/r05_phases_f32.txt
# the default line LAST: it reads profiles/r05_kernel_stats_replay_only.csv (copied there by the author from the run above) for frac_in_graph
cp $O/kernel_stats_replay_only.csv profiles/r05_kernel_stats_replay_only.csv 2> /dev/null
timeout -k 10 500 python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "default rc=$? at ${SECONDS}s"
ls -la $O

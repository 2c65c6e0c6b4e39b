#!/bin/bash
# round 4: final artefacts -- default line, rocprofv3 kernel statistics of the registration part of the default command, of the agent update
# and of the geometric update (160x512 and the C5 shape), phases.  Run through gpurun from the repository root.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
SECONDS=0
timeout -k 10 400 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "default rc=$? in ${SECONDS}s"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/reg --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-train-lines > $O/bench_register_only_under_rocprof.json 2> $O/reg.err ) && echo reg done
cp $O/reg/*/*kernel_stats.csv $O/kernel_stats_register_only.csv && rm -rf $O/reg
( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/tr --output-format csv -- python3 $R/bench.py --mode train --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_under_rocprof.json 2> $O/tr.err ) && echo train done
cp $O/tr/*/*kernel_stats.csv $O/kernel_stats_train.csv && rm -rf $O/tr
( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/tg --output-format csv -- python3 $R/bench.py --mode train-geo --num-pt 65536 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_geo_under_rocprof.json 2> $O/tg.err ) && echo geo done
cp $O/tg/*/*kernel_stats.csv $O/kernel_stats_train_geo.csv && rm -rf $O/tg
( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/c5 --output-format csv -- python3 $R/bench.py --mode train-geo --num-pt 65536 --img 352x1216 --prologue --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_geo_c5_under_rocprof.json 2> $O/c5.err ) && echo c5 done
cp $O/c5/*/*kernel_stats.csv $O/kernel_stats_train_geo_c5.csv && rm -rf $O/c5
timeout -k 10 300 python tools/phases.py sub > gpurun_out/r04_phases_f32.txt 2> /dev/null
cat gpurun_out/r04_phases_f32.txt
ls -la $O

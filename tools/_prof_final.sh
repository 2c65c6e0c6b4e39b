# Runs on the GPU box: rocprofv3 kernel statistics of the DEFAULT bench command (fp32, configs[1]) and of the bf16 mode, same step counts
# as bench.py's defaults, so that the average launch duration of the dominant kernel can be compared with the JSON line's avg_launch_us.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/f32 --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_f32_under_rocprof.json 2> $O/f32.err
echo f32 done
rocprofv3 --kernel-trace --stats -d $O/bf16 --output-format csv -- python3 $R/bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_under_rocprof.json 2> $O/bf16.err
echo bf16 done
cp $O/f32/*/*kernel_stats.csv $O/kernel_stats_f32.csv
cp $O/bf16/*/*kernel_stats.csv $O/kernel_stats_bf16.csv
rm -rf $O/f32 $O/bf16
ls -la $O

# GPU box: kernel trace of the replayed iteration (fp32 and bf16), then tools/critical.py + tools/trace_gaps.py on each
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in f32 bf16; do
  O=$R/gpurun_out/trace_$m
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace -d $O --output-format csv -- python3 $R/bench.py --dtype $m --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
  python3 $R/tools/critical.py $O 3 22 > $R/gpurun_out/crit_$m.txt 2>&1 || true
  python3 $R/tools/trace_gaps.py $O >> $R/gpurun_out/crit_$m.txt 2>&1 || true
  rm -rf $O
done

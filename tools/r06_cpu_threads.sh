#!/bin/bash
# round 6: the cpu_baseline leg (oracle, bounded sample of configs[1]) at 16 / 32 / 64 / all torch threads on the GPU box's
# host -> gpurun_out/r06_cpu_threads.txt.  No GPU work.
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06_cpu_threads.txt
echo "host: $(nproc) cpus visible, $(grep -c ^processor /proc/cpuinfo) in /proc/cpuinfo, $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)" > $out
for t in 16 32 64 $(nproc); do
  CMR_CPU_BASELINE_THREADS=$t timeout -k 10 400 python3 -c "
import json, os, bench
spec = json.load(open(os.path.join(bench.ROOT, 'tests', 'golden', 'specs.json')))
r = bench.cpu_baseline(spec)
print('threads %3d: %.3f registration iters/s   (%s)' % (r['cores'], r['value'], r['sample'].split('seconds per pass: ')[1]))
" 2>/dev/null | tee -a $out
done

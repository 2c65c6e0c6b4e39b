#!/bin/bash
# round 6: the driver's N > 1 command rehearsed on the one-GPU box -- every rank on device 0, gloo for the collectives.
# The pool's process guard allows at most 6 processes with the card open (a world-6 attempt was killed by it: 6 ranks + the elastic
# agent of torch.distributed.run = 7), so the card-side rehearsal is world 4; world 8 is rehearsed on the CPU
# (tests/test_launch_cpu.py::test_parent_starts_eight_ranks..., tests/test_dist_cpu.py world 4 / 8).  Throughput here is MEANINGLESS
# (four ranks time-slice one GPU); what the files prove: rendezvous, one graph capture per rank, distinct shard seeds, collective_ranks,
# the barrier / MAX protocol, ONE JSON line on stdout, exit code relayed.
# gpurun --timeout 900 -- 'bash tools/r06_rehearsal.sh'
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_rehearsal
rm -rf $O; mkdir -p $O
SECONDS=0
run() {  # run <tag> <seconds> <bench args...>
  tag=$1; lim=$2; shift 2
  timeout -k 10 $lim python bench.py "$@" --dist-backend gloo --share-gpu --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  rc=$?
  echo "$tag rc=$rc lines=$(wc -l < $O/$tag.json) at ${SECONDS}s"
  echo "{\"tag\": \"$tag\", \"rc\": $rc, \"stdout_lines\": $(wc -l < $O/$tag.json)}" >> $O/summary.jsonl
  return $rc
}
run register_w4 420 --gpus 4 --steps 3 --warmup 1 --no-pipeline-line --no-alone-pass &&
run train_w4 300 --gpus 4 --mode train --dtype bf16 --steps 5 --warmup 2 &&
run train_geo_w4 420 --gpus 4 --mode train-geo --steps 3 --warmup 1 &&
run train_geo_c5_w2 420 --gpus 2 --mode train-geo --num-pt 65536 --img 352x1216 --prologue --steps 2 --warmup 1
echo "done rc=$? at ${SECONDS}s"
tail -c 600 $O/*.err

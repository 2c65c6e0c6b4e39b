#!/bin/bash
# round 6 ("show with a replay-only A/B -- kernels removed from the graph -> step time -- what the row-GEMM family is worth"):
# bench.py --replay-only with every ops.linear launch (cmr_linear_f32 / cmr_linear_rows_bf16_f32: ~100 calls per registration) replaced by
# a ready-made zero map (CMR_AB_SKIP_LINEAR=1; results invalid, timing only), same box, alternating -> gpurun_out/r06_ab_skip_linear.txt
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_ab_skip_linear.txt
: > $O
for rep in 1 2; do
  for v in 0 1; do
    ms=$(CMR_AB_SKIP_LINEAR=$v timeout -k 10 200 python bench.py --replay-only --steps 20 --warmup 5 --no-cpu-baseline 2>>gpurun_out/r06_ab_skip_linear.err | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%.4f' % d['ms_per_step'])")
    echo "rep $rep  headline (fp32, configs[1])  row GEMMs skipped=$v -> ms_per_step $ms" | tee -a $O
  done
done

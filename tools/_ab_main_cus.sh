#!/bin/bash
# A/B on one box: C5 geometric update (352x1216, 65 536 points) with the image tower's persistent convolutions on all CUs vs a reservation of
# 16 / 32 CUs for the point chain that runs next to it (CMR_TAPE_MAIN_CUS, cmr_agent_amd/train/tape.py:Tape.MAIN_CUS)
out=gpurun_out/r05_ab_main_cus.txt; : > $out
for cus in 0 240 224 0 240 224; do
  CMR_TAPE_MAIN_CUS=$cus timeout -k 10 300 python bench.py --mode train-geo --num-pt 65536 --img 352x1216 --prologue --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/_ab.json 2> gpurun_out/_ab.err || { echo "FAILED $cus" >> $out; tail -3 gpurun_out/_ab.err >> $out; continue; }
  python - "$cus" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_ab.json").read().strip().split("\n")[-1])
print("image-tower CUs %-4s value %.2f  ms_per_step %.3f  loss %.6f" % (sys.argv[1] if sys.argv[1] != "0" else "all", d["value"], d["ms_per_step"], d["loss"]))
PY
done
cat $out

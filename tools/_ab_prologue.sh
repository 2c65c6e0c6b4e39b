#!/bin/bash
# A/B on one box: C5 geometric update with the point prologue one step ahead on a side stream vs in front of the step
out=gpurun_out/r04_ab_prologue.txt; : > $out
for flag in "--prologue-ahead" "" "--prologue-ahead" ""; do
  timeout -k 10 300 python bench.py --mode train-geo --num-pt 65536 --img 352x1216 --prologue $flag --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/_ab.json 2> gpurun_out/_ab.err || { echo "FAILED $flag" >> $out; continue; }
  python - "$flag" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_ab.json").read().strip().split("\n")[-1])
print("prologue %-18s value %.2f  ms_per_step %.3f  prologue_ms %.3f  loss %.6f" % (sys.argv[1] or "in front (default)", d["value"], d["ms_per_step"], d["prologue_ms"], d["loss"]))
PY
done
cat $out

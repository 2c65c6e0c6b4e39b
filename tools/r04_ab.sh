#!/bin/bash
# A/B of an environment switch on ONE box: tools/r04_ab.sh VAR "v1 v2 ..." <bench.py args> -> gpurun_out/r04_ab_<VAR>.txt
var=$1; vals=$2; shift 2
out=gpurun_out/r04_ab_${var}.txt
: > $out
for v in $vals $vals; do
  env $var=$v timeout -k 10 300 python bench.py "$@" > gpurun_out/_ab.json 2> gpurun_out/_ab.err || { echo "$var=$v FAILED" >> $out; continue; }
  python - "$var" "$v" >> $out <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_ab.json").read().strip().split("\n")[-1])
print("%s=%s  value %.2f  ms_per_step %.3f  launches %s" % (sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d.get("launches_per_step")))
PY
done
cat $out

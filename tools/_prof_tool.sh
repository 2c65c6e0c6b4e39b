#!/bin/bash
# rocprofv3 kernel statistics of one tools/*.py script: bash tools/_prof_tool.sh <tag> <script.py> [args...] -> gpurun_out/prof_<tag>_kernel_stats.csv
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
O=$R/gpurun_out/prof_$tag
rm -rf $O; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 "$R/$1" "${@:2}" > $O.out 2> $O.err )
cp $O/*/*kernel_stats.csv $R/gpurun_out/prof_${tag}_kernel_stats.csv && rm -rf $O
python3 - "$R/gpurun_out/prof_${tag}_kernel_stats.csv" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:25]:
    print("%-100s calls %5s  avg %9.1f us  total %8.2f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P

# ordered kernel trace of ONE eager registration iteration (who launches the copy kernels?) -> gpurun_out/trace_eager/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trace_eager
rm -rf $O; mkdir -p $O
CMR_STREAMS=0 rocprofv3 --kernel-trace -d $O --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline > $O/bench.json 2> $O/bench.err
F=$(find $O -name "*kernel_trace.csv" | head -1)
python3 - "$F" > $O/sequence.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    print(r["Kernel_Name"][:90], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
PY
wc -l $O/sequence.txt

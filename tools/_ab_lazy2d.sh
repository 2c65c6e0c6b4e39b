cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ab2d; rm -rf $O; mkdir -p $O
for v in 0 1; do
  export CMR_AGENT_LAZY_2D=$v
  rocprofv3 --kernel-trace --stats -d $O/v$v --output-format csv -- python3 $R/bench.py --mode train --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/v$v.json 2> $O/v$v.err
  cp $O/v$v/*/*kernel_stats.csv $O/stats_$v.csv
done
python3 - <<PY
import csv
for v in (0,1):
    rows=list(csv.DictReader(open("$O/stats_%d.csv"%v)))
    tot=sum(float(r["TotalDurationNs"]) for r in rows)
    print("lazy2d=%d total kernel ms %.2f"%(v,tot/1e6))
    for r in rows:
        n=r["Name"]
        if any(k in n for k in ("mm_kernel","wgrad_bf16_tr","affine_act","bf16_tt","bn_stats_partial")):
            print("   %-70s %5s x %7.1f us = %6.2f ms"%(n.replace("(anonymous namespace)::","")[:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY

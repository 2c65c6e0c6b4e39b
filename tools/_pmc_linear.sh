# PMC passes over the fp32 row-map GEMM alone (tools/prof_linear.py): where do the waves' cycles go
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_linear
rm -rf $O; mkdir -p $O
for v in 1 0; do
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/p$v --output-format csv -- python3 $R/tools/prof_linear.py 524288 64 $v > $O/p$v.log 2>&1
done
python3 - <<EOF2
import csv, glob, collections
for v in (1, 0):
    f = glob.glob("$O/p%d/*/*counter_collection.csv" % v)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "linear_w" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, c in agg.items():
        n = 6.0
        wc = c["SQ_WAVE_CYCLES"]
        print(k[:60], {a: round(b / n) for a, b in c.items()})
        print("   share of wave (quad-)cycles: wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f ; mfma busy cycles / (GUI_ACTIVE/8 * 1024) = %.3f" % (
            c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
EOF2

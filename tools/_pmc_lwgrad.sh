# PMC passes over the LDS-staged row-map weight gradient alone (tools/prof_lwgrad.py): where do the waves' cycles go
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_lwgrad
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d $O/p1 --output-format csv -- python3 $R/tools/prof_lwgrad.py > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/p2 --output-format csv -- python3 $R/tools/prof_lwgrad.py > $O/p2.log 2>&1
python3 - <<EOF2
import csv, glob, collections
for v in (1, 2):
    f = glob.glob("$O/p%d/*/*counter_collection.csv" % v)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "linear_wgrad" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, c in agg.items():
        print(k[:70], {a: round(b / 6.0) for a, b in c.items()})
        wc = c["SQ_WAVE_CYCLES"]
        if v == 1 and wc:
            print("   share of wave cycles: wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f  active_valu %.2f active_lds %.2f; mfma busy = %.3f" % (
                c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_ACTIVE_INST_LDS"] / wc,
                c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
EOF2

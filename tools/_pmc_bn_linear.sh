# PMC passes over the fused linear + BatchNorm layer kernels alone (tools/prof_bn_linear.py): where do the waves' cycles go
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_bn_linear
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/p1 --output-format csv -- python3 $R/tools/prof_bn_linear.py > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p2 --output-format csv -- python3 $R/tools/prof_bn_linear.py > $O/p2.log 2>&1
python3 - <<EOF2
import csv, glob, collections
for v in (1, 2):
    fs = glob.glob("$O/p%d/*/*counter_collection.csv" % v)
    if not fs:
        print("pass", v, "no counters"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); ns = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("bn_linear", "bn_bwd_partial", "affine_act")): continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); cnt[k] += 1; ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for k, c in agg.items():
        n = float(cnt[k]); wc = c["SQ_WAVE_CYCLES"]
        print(k[:110], "x%d  %.1f us" % (n, ns[k] / n / 1e3))
        print("    ", {a: round(b / n) for a, b in c.items()})
        if v == 1:
            print("     share of wave cycles: wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f ; mfma busy / (GUI_ACTIVE/8 * 1024) = %.3f" % (
                c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
        else:
            print("     LDS: active %.2f of wave cycles, bank conflict cycles / active %.2f, wait_inst_lds %.2f" % (
                c["SQ_ACTIVE_INST_LDS"] / wc, c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_ACTIVE_INST_LDS"], 1), c["SQ_WAIT_INST_LDS"] / wc))
EOF2

# PMC passes over the bf16 3x3 weight gradient alone (tools/prof_wgrad_bf16.py): HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and
# where the waves' cycles go -> gpurun_out/pmc_wgrad_bf16/summary.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_wgrad_bf16
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/tools/prof_wgrad_bf16.py > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/tools/prof_wgrad_bf16.py > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/p1 --output-format csv -- python3 $R/tools/prof_wgrad_bf16.py > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p2 --output-format csv -- python3 $R/tools/prof_wgrad_bf16.py > $O/p2.log 2>&1
python3 - > $O/summary.txt <<EOF2
import csv, glob, collections
def load(tag):
    fs = glob.glob("$O/%s/*/*counter_collection.csv" % tag)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); ns = collections.defaultdict(float); seen = set()
    for r in csv.DictReader(open(fs[0])) if fs else []:
        k = r["Kernel_Name"]
        if "wgrad" not in k and "bias_reduce" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); cnt[k] += 1; ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, cnt, ns
# FETCH_SIZE / WRITE_SIZE: KiB units; FETCH_SIZE is doubled on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)
for tag, name, mul in (("fetch", "FETCH_SIZE", 2.0 * 1024), ("write", "WRITE_SIZE", 1024.0)):
    agg, cnt, ns = load(tag)
    for k, c in agg.items():
        print("%-8s %-90s x%d  %.1f us  %.1f MB per launch" % (tag, k[:90], cnt[k], ns[k] / cnt[k] / 1e3, c[name] * mul / cnt[k] / 1e6))
for v in (1, 2):
    agg, cnt, ns = load("p%d" % v)
    for k, c in agg.items():
        n = float(cnt[k]); wc = c["SQ_WAVE_CYCLES"]
        print(k[:110], "x%d  %.1f us" % (n, ns[k] / n / 1e3))
        if v == 1:
            print("     share of wave cycles: wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f  vmem %.2f  valu %.2f; mfma busy / (GUI_ACTIVE/8 * 1024) = %.3f" % (
                c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_ACTIVE_INST_VMEM"] / wc, c["SQ_ACTIVE_INST_VALU"] / wc,
                c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
        else:
            print("     LDS: active %.2f of wave cycles, bank conflict cycles / active %.2f, wait_inst_lds %.2f; LDS instructions %.0f, VALU %.0f per wave-cycle-k" % (
                c["SQ_ACTIVE_INST_LDS"] / wc, c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_ACTIVE_INST_LDS"], 1), c["SQ_WAIT_INST_LDS"] / wc, c["SQ_INSTS_LDS"] / n, c["SQ_INSTS_VALU"] / n))
EOF2
cat $O/summary.txt

# Where do the waves of the Winograd kernel spend their cycles?  Two PMC passes over tools/prof_wino.py at two shapes (the agent's
# 88x304 128 -> 128 map and the full-resolution 352x1216 64 -> 64 map): wave-cycle shares (waiting on anything / on instruction issue /
# active), MFMA busy, VALU and VMEM activity; LDS activity, bank conflicts, waits on LDS.  -> gpurun_out/pmc_wino_waits.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_wino_waits
rm -rf $O; mkdir -p $O
for shp in "88 304 128 128" "352 1216 64 64"; do
  tag=$(echo $shp | tr ' ' 'x')
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $O/a_$tag --output-format csv -- python3 $R/tools/prof_wino.py $shp > $O/a_$tag.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/b_$tag --output-format csv -- python3 $R/tools/prof_wino.py $shp > $O/b_$tag.log 2>&1
done
python3 - <<EOF2
import csv, glob, collections
for tag in ("88x304x128x128", "352x1216x64x64"):
    for v in ("a", "b"):
        fs = glob.glob("$O/%s_%s/*/*counter_collection.csv" % (v, tag))
        if not fs:
            print(tag, v, "no counters"); continue
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); ns = collections.defaultdict(float); seen = set()
        for r in csv.DictReader(open(fs[0])):
            k = r["Kernel_Name"]
            if "wino" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); cnt[k] += 1; ns[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        for k, c in agg.items():
            n = float(cnt[k]); wc = c["SQ_WAVE_CYCLES"]
            print(tag, k.replace("(anonymous namespace)::", "")[:70], "x%d  %.1f us" % (n, ns[k] / n / 1e3))
            if v == "a":
                print("     share of wave cycles: wait_any %.2f  wait_inst_any %.2f  active_inst_any %.2f  active VALU %.2f  active VMEM %.2f ; mfma busy / (GUI_ACTIVE/8 * 1024) = %.3f" % (
                    c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_ACTIVE_INST_VMEM"] / wc,
                    c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)))
            else:
                print("     LDS: active %.2f of wave cycles, bank conflict cycles / active %.2f, wait_inst_lds %.2f ; VALU instructions per wave cycle %.3f" % (
                    c["SQ_ACTIVE_INST_LDS"] / wc, c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_ACTIVE_INST_LDS"], 1), c["SQ_WAIT_INST_LDS"] / wc, c["SQ_INSTS_VALU"] / wc))
EOF2

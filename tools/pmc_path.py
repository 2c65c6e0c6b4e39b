"""gpurun_out/pmc_path/{fetch,write,sq}/**/counter_collection.csv -> per-kernel summary (JSON on stdout): launches, mean
duration under the profiler, HBM bytes per launch (FETCH_SIZE doubled on gfx950, KiB units; MI355X_MICROARCH.md HBM
section), matrix-pipe duty = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), effective clock.
Only the LAST third of the dispatches of every kernel is used (the profiled command runs warm-up + timed + timing pass)."""
import collections
import csv
import glob
import json
import os
import re
import sys

root = sys.argv[1]


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    m = re.match(r"(void )?([A-Za-z0-9_:]+(<[^>]*>)?)", n)
    return m.group(2) if m else n[:60]


def collect(sub):
    files = sorted(glob.glob(root + "/" + sub + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    per = collections.defaultdict(lambda: collections.OrderedDict())      # kernel -> dispatch -> {counter: value, _ns}
    for r in csv.DictReader(open(files[-1])):
        k = short(r["Kernel_Name"])
        d = per[k].setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    out = {}
    for k, disp in per.items():
        ids = sorted(disp)
        ids = ids[len(ids) * 2 // 3:] if len(ids) >= 3 else ids
        n = len(ids)
        agg = collections.defaultdict(float)
        for i in ids:
            for c, v in disp[i].items():
                agg[c] += v
        out[k] = {c: v / n for c, v in agg.items()}
        out[k]["_launches"] = n
    return out


fe, wr, sq = collect("fetch"), collect("write"), collect("sq")
res = {"_note": "rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES "
                "GRBM_GUI_ACTIVE) of `CMR_STREAMS=0 bench.py --steps 1 --warmup 1 --eager` (tools/_pmc_path.sh); per-launch means over the "
                "last iteration; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled (gfx950); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
                "(GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); clock = GRBM_GUI_ACTIVE / 8 / duration (reads high below ~0.3 ms)"}
for k in sorted(sq, key=lambda k: -sq[k]["_ns"] * sq[k]["_launches"]):
    if k.startswith("at::") or "rocclr" in k or k not in fe or k not in wr:
        continue
    s = sq[k]
    cyc = s.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    res[k] = {"launches_per_iteration": s["_launches"], "mean_us_under_pmc": round(s["_ns"] / 1e3, 2),
              "hbm_fetch_bytes": round(2 * 1024 * fe[k].get("FETCH_SIZE", 0.0)), "hbm_write_bytes": round(1024 * wr[k].get("WRITE_SIZE", 0.0)),
              "mfma_busy": round(s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4) if cyc else None,
              "clock_ghz": round(cyc / s["_ns"], 3) if s["_ns"] else None, "waves": round(s.get("SQ_WAVES", 0.0))}
print(json.dumps(res, indent=1))

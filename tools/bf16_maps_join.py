"""gpurun_out/bf16_maps_<w>.json (tools/bf16_maps.py, HIP-event times alone) + the FETCH_SIZE / WRITE_SIZE passes of the same program
(gpurun_out/pmc_bf16_<w>/{fetch,write}) -> the bf16 3x3 convolutions per MAP SIZE: launches per iteration, microseconds alone, algorithmic and
counted HBM bytes per launch, GB/s and the fraction of 8 TB/s (JSON on stdout).  Launches are joined by order: the last N
dispatches of conv3x3_bf16* in a counter pass are the N logged calls of the last iteration (one kernel per call, one stream).
FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys

calls_path, pmc_root = sys.argv[1], sys.argv[2]
log = json.load(open(calls_path))
calls = log["calls"]
n = len(calls)


def counter(sub, name):
    files = sorted(glob.glob(pmc_root + "/" + sub + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not files:
        return None, None
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(files[-1])):
        if "conv3x3_bf16" not in r["Kernel_Name"]:
            continue
        d = disp.setdefault(int(r["Dispatch_Id"]), dict(v=0.0, k=r["Kernel_Name"], grid=int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)))
        if r["Counter_Name"] == name:
            d["v"] += float(r["Counter_Value"])
    ids = sorted(disp)[-n:]
    if len(ids) != n:
        raise SystemExit("%s: %d conv3x3_bf16 dispatches, %d calls logged" % (sub, len(ids), n))
    return [disp[i]["v"] for i in ids], [(disp[i]["k"], disp[i]["grid"]) for i in ids]


fe, kern = counter("fetch", "FETCH_SIZE")
wr, _ = counter("write", "WRITE_SIZE")
rows = collections.OrderedDict()
for i, c in enumerate(calls):
    key = (c["H"], c["W"], c["Cin"], c["Cout"], c["stride"], c["pool"], c["x_bf16"], c["y_bf16"], c["res"], c["B"])
    r = rows.setdefault(key, dict(n=0, us=0.0, by=0.0, fl=0.0, hbm=0.0, kernel=None, wgs=None))
    r["n"] += 1
    r["us"] += c["us"]
    r["by"] += c["bytes"]
    r["fl"] += c["flops"]
    if fe is not None and wr is not None:
        r["hbm"] += 2 * 1024 * fe[i] + 1024 * wr[i]
        k = kern[i][0].replace("(anonymous namespace)::", "").replace("void ", "")
        r["kernel"], r["wgs"] = k[:k.find("(")] if "(" in k else k, kern[i][1]
tot_us, tot_by, tot_hbm = sum(r["us"] for r in rows.values()), sum(r["by"] for r in rows.values()), sum(r["hbm"] for r in rows.values())
out = dict(_note="bf16 3x3 convolutions of one eager registration iteration of %s, every branch on one stream; per map size: launches, mean "
                 "microseconds ALONE (HIP events), algorithmic bytes and PMC-counted HBM bytes per launch (FETCH_SIZE x 2 x 1 KiB + WRITE_SIZE x "
                 "1 KiB, separate passes), achieved = algorithmic bytes / time; frac = achieved / 8 TB/s" % log["name"],
           workload=log["workload"], launches_per_iteration=n, conv_ms_alone_per_iteration=tot_us / 1e3,
           algorithmic_mb_per_iteration=tot_by / 1e6, hbm_mb_per_iteration=tot_hbm / 1e6 if tot_hbm else None,
           frac_blended=tot_by / (tot_us * 1e-6) / 8e12, traffic_over_algorithmic=(tot_hbm / tot_by) if tot_hbm else None,
           algorithmic_bytes_per_launch=tot_by / n, hbm_bytes_per_launch=(tot_hbm / n) if tot_hbm else None, maps=[])
for key, r in sorted(rows.items(), key=lambda kv: -kv[1]["us"]):
    H, W, ci, co, s, pool, xb, yb, res, B = key
    m = r["n"]
    out["maps"].append(dict(map="%dx%d" % (H, W), B=B, cin=ci, cout=co, stride=s, pool=pool, x_bf16=xb, y_bf16=yb, residual=res, launches=m,
                            us_alone=round(r["us"] / m, 2), share_of_conv_time=round(r["us"] / tot_us, 4),
                            algorithmic_mb=round(r["by"] / m / 1e6, 3), hbm_mb=round(r["hbm"] / m / 1e6, 3) if r["hbm"] else None,
                            traffic_over_algorithmic=round(r["hbm"] / r["by"], 3) if r["hbm"] else None,
                            gbs=round(r["by"] / (r["us"] * 1e-6) / 1e9, 1), frac_hbm=round(r["by"] / (r["us"] * 1e-6) / 8e12, 4),
                            tflops=round(r["fl"] / (r["us"] * 1e-6) / 1e12, 1), frac_bf16_mfma=round(r["fl"] / (r["us"] * 1e-6) / 2500e12, 4),
                            kernel=r["kernel"], workgroups=r["wgs"]))
print(json.dumps(out, indent=1))

"""Sums rocprofv3 counter_collection.csv per counter for kernels matching a substring (last dispatch only)."""
import csv, glob, sys, collections
pat = sys.argv[2] if len(sys.argv) > 2 else "wino"
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.OrderedDict(); last = None
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        d = r["Dispatch_Id"]
        acc.setdefault(d, collections.OrderedDict())
        acc[d][r["Counter_Name"]] = acc[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        acc[d]["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        acc[d]["_res"] = "lds=%s scratch=%s vgpr=%s agpr=%s" % (r["LDS_Block_Size"], r["Scratch_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"])
    if acc:
        d = list(acc)[-1]
        print(f.split("/")[-3], {k: (v if isinstance(v, str) else round(v)) for k, v in acc[d].items()})

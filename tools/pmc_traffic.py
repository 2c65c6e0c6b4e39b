"""gpurun_out/prof_r/{fetch,write}/**/counter_collection.csv -> per-kernel HBM bytes per launch (JSON on stdout).
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled on gfx950 (128-B requests tallied at 64 B, MI355X_MICROARCH.md)."""
import csv, glob, json, collections, os, re, sys
root = sys.argv[1]
def collect(sub, counter):
    acc = collections.defaultdict(lambda: [set(), 0.0])
    files = sorted(glob.glob(root + "/" + sub + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for f in files[-1:]:                                  # gpurun merges runs into the same directory: newest only
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"].replace("(anonymous namespace)::", "")
            m = re.match(r"(void )?([A-Za-z0-9_:]+(<[^>]*>)?)", n)
            k = m.group(2) if m else n[:60]
            acc[k][0].add(r["Dispatch_Id"]); acc[k][1] += float(r["Counter_Value"])
    return acc
fe, wr = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate runs of `bench.py --steps 1 --warmup 1 --eager`; "
                "counters are KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); "
                "averages over all launches of the kernel"}
for k in sorted(fe, key=lambda k: -fe[k][1]):
    if k.startswith("at::") or "rocclr" in k or k not in wr:
        continue
    nf, nw = len(fe[k][0]), len(wr[k][0])
    f, w = 2 * 1024 * fe[k][1] / nf, 1024 * wr[k][1] / nw
    out[k] = {"launches": nf, "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w}
print(json.dumps(out, indent=1))

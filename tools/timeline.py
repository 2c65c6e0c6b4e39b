"""Time-ordered kernel list of the last hipGraph-replayed iteration of a rocprofv3 kernel trace of bench.py (start offset, duration, gap to
the previous END on the device, kernel) -> stdout.  python tools/timeline.py <trace dir>"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")) for r in rows)
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:60]
stems = [i for i, e in enumerate(ev) if "stem_a_kernel" in e[2]]
a, b = stems[-2], stems[-1]
seg = ev[a:b]
t0 = seg[0][0]
cur_e = seg[0][0]
for s, e, n, g, wg in seg:
    print("%9.1f %8.1f %7.1f  %-60s %s/%s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - cur_e) / 1e3, short(n), g, wg))
    cur_e = max(cur_e, e)

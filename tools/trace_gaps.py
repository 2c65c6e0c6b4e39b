"""Last hipGraph-replayed iteration of a rocprofv3 kernel trace of bench.py: device idle time (no kernel running), busiest kernels,
and the timeline of the agent loop (observation -> 2-D / 3-D embedding -> heads -> pose step)."""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:48]
# iterations are delimited by the stem kernel (one per geo forward)
stems = [i for i, e in enumerate(ev) if "stem_a_kernel" in e[2]]
a, b = stems[-2], stems[-1]
seg = ev[a:b]
t0, t1 = seg[0][0], max(e[1] for e in seg)
# union of busy intervals
busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
for s, e, _ in seg[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("iteration span %.2f ms, device busy (union of kernel intervals) %.2f ms, idle %.2f ms, sum of kernel durations %.2f ms, kernels %d" % (
    (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, sum(e - s for s, e, _ in seg) / 1e6, len(seg)))
# idle gaps > 3 us: what ran before / after
gaps = []
cur_e, last = seg[0][1], seg[0][2]
for s, e, n in seg[1:]:
    if s > cur_e + 3000:
        gaps.append((s - cur_e, short(last), short(n)))
    if e > cur_e:
        cur_e, last = e, n
gaps.sort(reverse=True)
print("idle gaps > 3 us: %d, total %.2f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e6))
for g in gaps[:15]:
    print("  %.1f us after %-48s before %s" % (g[0] / 1e3, g[1], g[2]))

"""Per-segment kernel breakdown of a rocprofv3 kernel trace of bench.py: one agent step and the geo forward
(the span between the last pose_step of one iteration and the first of the next)."""
import csv, glob, collections, re, sys
import os
f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z0-9_:]+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or ''))[:46]
idx = [i for i, r in enumerate(rows) if 'pose_step' in r['Kernel_Name']]
segs = list(zip(idx[:-1], idx[1:]))
big = [s for s in segs if s[1] - s[0] > 100]
small = [s for s in segs if s[1] - s[0] <= 100]
for name, (a, b) in (("agent step", small[len(small) // 2]), ("geo forward + first agent step", big[len(big) // 2])):
    d = collections.OrderedDict()
    for r in rows[a + 1:b + 1]:
        k = short(r['Kernel_Name'])
        d.setdefault(k, [0, 0.0]); d[k][0] += 1; d[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot = sum(v[1] for v in d.values())
    print("%s: %d kernels, %.2f ms" % (name, sum(v[0] for v in d.values()), tot / 1e3))
    for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
        print("  %-48s x%4d %9.1f us  %5.1f%%" % (k, v[0], v[1], 100 * v[1] / tot))

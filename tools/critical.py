"""Timeline analysis of one replayed registration iteration from a rocprofv3 kernel trace: wall time, time with
exactly one kernel type resident (where that kernel alone is on the critical path), idle time."""
import csv, glob, collections, re, sys
import os
f = max(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z0-9_:]+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or ''))[:44]
idx = [i for i, r in enumerate(rows) if 'pose_step' in r['Kernel_Name']]
# iteration = 10 pose_steps; pick the iteration in the middle of the replayed ones (2nd..4th)
its = [idx[i:i + 10] for i in range(0, len(idx) - 9, 10)]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = its[k - 1][-1] + 1, its[k][-1] + 1
seg = rows[a:b]
ev = []
for r in seg:
    ev.append((int(r['Start_Timestamp']), 1, short(r['Kernel_Name'])))
    ev.append((int(r['End_Timestamp']), -1, short(r['Kernel_Name'])))
ev.sort(key=lambda e: (e[0], e[1]))
active = collections.Counter(); last = ev[0][0]; solo = collections.Counter(); idle = 0; multi = 0
for t, d, name in ev:
    dt = t - last
    n = sum(active.values())
    if dt > 0:
        if n == 0: idle += dt
        elif n == 1: solo[[x for x in active if active[x] > 0][0]] += dt
        else: multi += dt
    active[name] += d; last = t
wall = ev[-1][0] - ev[0][0]
print("iteration %d: %d kernels, wall %.2f ms, idle %.2f ms, >=2 kernels resident %.2f ms" % (k, len(seg), wall / 1e6, idle / 1e6, multi / 1e6))
for name, t in solo.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
    print("  solo %-46s %8.2f ms" % (name, t / 1e6))

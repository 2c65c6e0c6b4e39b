"""Timeline of ONE replayed agent update from a rocprofv3 kernel trace of `bench.py --mode train` (steps delimited by adam_kernel): per queue
busy time and gaps, time with 0 / 1 / >= 2 kernels resident, the kernels that run alone, and the ordered kernel list.
python tools/train_timeline.py <trace dir> [step index from the end, default 2] [marker kernel, default adam_kernel; stem_a_kernel for a
registration trace] > timeline.txt"""
import collections
import csv
import glob
import os
import re
import sys

f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'([A-Za-z0-9_:]+)(<[^>]*>)?', n)
    return (m.group(1) + (m.group(2) or ''))[:60]


marker = sys.argv[3] if len(sys.argv) > 3 else 'adam_kernel'      # a step ENDS with adam_kernel; any other marker (stem_a_kernel) STARTS one
marks = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = (marks[-k - 1] + 1, marks[-k] + 1) if marker == 'adam_kernel' else (marks[-k - 1], marks[-k])
seg = rows[a:b]
t0 = int(seg[0]['Start_Timestamp'])
wall = max(int(r['End_Timestamp']) for r in seg) - t0
qkey = 'Queue_Id' if 'Queue_Id' in seg[0] else 'Stream_Id'
ev = []
for r in seg:
    ev.append((int(r['Start_Timestamp']), 1, short(r['Kernel_Name'])))
    ev.append((int(r['End_Timestamp']), -1, short(r['Kernel_Name'])))
ev.sort(key=lambda e: (e[0], e[1]))
active = collections.Counter()
last = ev[0][0]
solo = collections.Counter()
idle = multi = 0
for t, d, name in ev:
    dt = t - last
    n = sum(active.values())
    if dt > 0:
        if n == 0:
            idle += dt
        elif n == 1:
            solo[[x for x in active if active[x] > 0][0]] += dt
        else:
            multi += dt
    active[name] += d
    last = t
print("step: %d kernels, wall %.3f ms, nothing resident %.3f ms, exactly one kernel %.3f ms, >= 2 kernels %.3f ms" % (
    len(seg), wall / 1e6, idle / 1e6, sum(solo.values()) / 1e6, multi / 1e6))
byq = collections.defaultdict(list)
for r in seg:
    byq[r[qkey]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    gaps = [int(rs[i + 1]['Start_Timestamp']) - int(rs[i]['End_Timestamp']) for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g > 0]
    print("queue %s: %d kernels, busy %.3f ms, from %.3f to %.3f ms, %d gaps summing %.3f ms (median %.1f us)" % (
        q, len(rs), busy / 1e6, (int(rs[0]['Start_Timestamp']) - t0) / 1e6, (int(rs[-1]['End_Timestamp']) - t0) / 1e6, len(pos), sum(pos) / 1e6,
        sorted(pos)[len(pos) // 2] / 1e3 if pos else 0.0))
print("kernels that run ALONE on the device (time with exactly this one resident):")
for name, t in solo.most_common(25):
    print("  %-62s %7.3f ms" % (name, t / 1e6))
print("ordered:")
for r in seg:
    print("  %8.1f us  +%7.1f us  q%-3s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r[qkey],
                                              short(r['Kernel_Name'])))

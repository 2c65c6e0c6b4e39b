"""python tools/side_queue_probe4_parse.py <trace dir>: per replay of tools/side_queue_probe4.py -- start of the side branch's first kernel
after the origin's first kernel ended, end of the side chain, end of the origin chain (microseconds after the replay's first kernel)."""
import csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
cfgs = [(int(x.split(":")[0]), x.split(":")[1]) for x in os.environ.get("PROBE4", "").split(",") if x] or [
    (n, o) for n in (10, 40, 80, 160, 320) for o in ("main_first", "side_first", "interleaved")]
reps, cur = [], None
for r in rows:
    n = r['Kernel_Name']
    if 'sigmoid' in n:
        cur = []
    if cur is not None:
        cur.append(r)
    if 'cos' in n.lower() and 'sigmoid' not in n and cur is not None and len(cur) > 3:
        reps.append(cur)
        cur = None
reps = [r for r in reps if sum('CUDAFunctor_add' in x['Kernel_Name'] or 'AddFunctor' in x['Kernel_Name'] or 'add' in x['Kernel_Name'].lower() for x in r) >= 30]
print("%d replays found" % len(reps))
for i, rep in enumerate(reps):
    t0 = int(rep[0]['Start_Timestamp'])
    muls = [x for x in rep if 'Mul' in x['Kernel_Name'] or 'mul' in x['Kernel_Name']]
    adds = [x for x in rep if x not in muls and ('add' in x['Kernel_Name'].lower())]
    if not muls or not adds:
        continue
    dep_end = int(muls[0]['End_Timestamp'])
    cfg = cfgs[(i // 3) % len(cfgs)] if len(reps) == 3 * len(cfgs) else ("?", "?")
    print("n_main %4s %-12s rep %d: side starts %7.1f us after its dependency ended; side chain (%d) ends at %7.1f us, origin chain (%d) at %7.1f us" % (
        cfg[0], cfg[1], i % 3, (int(adds[0]['Start_Timestamp']) - dep_end) / 1e3, len(adds), (int(adds[-1]['End_Timestamp']) - t0) / 1e3, len(muls),
        (int(muls[-1]['End_Timestamp']) - t0) / 1e3))

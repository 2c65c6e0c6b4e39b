"""Kernel census of ONE replayed registration step from a rocprofv3 kernel trace of `bench.py --replay-only` (steps delimited by stem_a_kernel):
dispatches per kernel name in the last replay, and the __amd_rocclr_copyBuffer dispatches of the whole trace split into [before the first
replay = model load / plan building / capture] and [per replay].  python tools/replay_census.py <trace dir> [marker substring]"""
import collections
import csv
import glob
import os
import re
import sys

f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
mark = sys.argv[2] if len(sys.argv) > 2 else "stem_a_kernel"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:70]
marks = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
copies = [i for i, r in enumerate(rows) if "copyBuffer" in r["Kernel_Name"]]
print("%d dispatches, %d x %s, %d x __amd_rocclr_copyBuffer in the whole trace" % (len(rows), len(marks), mark, len(copies)))
print("copyBuffer dispatches before the first %s (state_dict upload, plans): %d" % (mark, sum(1 for i in copies if i < marks[0])))
for j in range(len(marks)):
    a, b = marks[j], marks[j + 1] if j + 1 < len(marks) else len(rows)
    seg = rows[a:b]
    n = sum(1 for r in seg if "copyBuffer" in r["Kernel_Name"])
    wall = (max(int(r["End_Timestamp"]) for r in seg) - int(seg[0]["Start_Timestamp"])) / 1e6
    print("  step %2d: %4d dispatches, %2d copyBuffer, %.3f ms from first start to last end" % (j, len(seg), n, wall))
a, b = marks[-2], marks[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
print("copyBuffer dispatches of the last full step (offset, duration, previous kernel -> next kernel on the device):")
for i, r in enumerate(seg):
    if "copyBuffer" in r["Kernel_Name"]:
        print("  %9.1f us +%5.1f us   after %-40s before %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                            short(seg[i - 1]["Kernel_Name"])[:40] if i else "-", short(seg[i + 1]["Kernel_Name"])[:40] if i + 1 < len(seg) else "-"))
cnt, dur = collections.Counter(), collections.Counter()
for r in seg:
    cnt[short(r["Kernel_Name"])] += 1
    dur[short(r["Kernel_Name"])] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print("kernels of the last full step:")
for k, v in sorted(dur.items(), key=lambda kv: -kv[1]):
    print("  %5d x %8.1f us = %8.3f ms  %s" % (cnt[k], v / cnt[k] / 1e3, v / 1e6, k))

"""Launch-by-launch timeline of ONE steady-state step out of a rocprofv3 kernel_trace.csv: per hardware queue, every
kernel in start order with its offset from the step's start, its duration and the idle gap before it.
Usage: python tools/step_timeline.py <dir with */*_kernel_trace.csv> [min_us]"""
import csv
import glob
import sys
from collections import defaultdict

path = sys.argv[1]
f = glob.glob(path + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fps_pruned_kernel" in r["Kernel_Name"]]
pairs = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if marks[i + 1] - marks[i] > 200]
a, b = pairs[-1]
win = rows[a:b]
t0 = int(win[0]["Start_Timestamp"])
queues = defaultdict(list)
for r in win:
    queues[r.get("Queue_Id", "0")].append(r)
for q, rs in sorted(queues.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"== queue {q}: {len(rs)} launches, busy {1e-6 * busy:.3f} ms")
    prev_end = None
    for r in rs:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = 0 if prev_end is None else s - prev_end
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{1e-3 * (s - t0):9.1f} us  {1e-3 * (e - s):7.1f} us  gap {1e-3 * gap:6.1f}  {name[:110]}")
        prev_end = e if prev_end is None else max(prev_end, e)

"""Steady-state step of a graph-mode bench trace: main-path vs side-stream (geometry) kernels, idle gaps."""
import csv, glob, sys
from collections import defaultdict
f = (glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv") + glob.glob(sys.argv[1] + "/*_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = "fps_pruned_kernel" if any("fps_pruned_kernel" in r["Kernel_Name"] for r in rows) else "fps_kernel<1024"
marks = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
pairs = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if marks[i + 1] - marks[i] > 500]
a, b = pairs[-1]
win = rows[a:b]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
side_names = ("fps_", "ball_query", "three_nn", "gather_points_kernel")
# vote-aggregation geometry runs inline on the main stream; approximate the split by stream/queue id when present
key = "Queue_Id" if "Queue_Id" in rows[0] else None
side = [r for r in win if any(s in r["Kernel_Name"] for s in side_names)]
main = [r for r in win if r not in side]
def union(rs):
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rs)
    tot, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs, iv[0][0], iv[-1][1]
wall = int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])
mu, m0, m1 = union(main)
print(f"step wall {wall/1e6:.3f} ms | main path: {len(main)} launches, sum {sum(dur(r) for r in main)/1e6:.3f} ms, union busy {mu/1e6:.3f} ms, span {(m1-m0)/1e6:.3f} ms")
print(f"side (geometry) kernels: {len(side)} launches, sum {sum(dur(r) for r in side)/1e6:.3f} ms")
agg = defaultdict(lambda: [0, 0])
for r in main: agg[r["Kernel_Name"]][0] += dur(r); agg[r["Kernel_Name"]][1] += 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:n]:
    print(f"{d/1e6:8.3f} ms {c:5d}x avg {d/c/1e3:8.1f} us  {k[:110]}")
small = [r for r in main if dur(r) < 10000]
print(f"main-path kernels < 10 us: {len(small)} launches, {sum(dur(r) for r in small)/1e6:.3f} ms")

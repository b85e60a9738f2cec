"""Summarise ONE steady-state step out of a rocprofv3 kernel_trace.csv (warm-up excluded):
the window between the last two launches of the SA1 FPS kernel (one per step)."""
import csv
import glob
import sys
from collections import defaultdict

path = sys.argv[1]
f = glob.glob(path + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fps_pruned_kernel" in r["Kernel_Name"] or "fps_pruned_reg_kernel" in r["Kernel_Name"]
         or "fps_kernel<1024" in r["Kernel_Name"]]
pairs = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if marks[i + 1] - marks[i] > 200]
a, b = pairs[-1]  # the last FULL step (bench.py launches the FPS kernel alone again in its per-kernel section)
win = rows[a:b]
t0, t1 = int(win[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
agg = defaultdict(lambda: [0, 0])
for r in win:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[r["Kernel_Name"]][0] += d
    agg[r["Kernel_Name"]][1] += 1
busy = sum(v[0] for v in agg.values())
print(f"step wall {1e-6 * (t1 - t0):.3f} ms, kernel busy {1e-6 * busy:.3f} ms, {len(win)} launches")
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for k, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{1e-6 * d:8.3f} ms {n:5d}x  {k[:120]}")

# per-queue view of the same step: span and busy time of every HIP queue, and (with a third argument) the launches in order
qs = defaultdict(list)
for r in win:
    qs[r["Queue_Id"]].append(r)
print("\nqueues of this step (offsets from the step's first launch):")
for q, rs in sorted(qs.items(), key=lambda kv: -len(kv[1])):
    s = min(int(r["Start_Timestamp"]) for r in rs) - t0
    e = max(int(r["End_Timestamp"]) for r in rs) - t0
    bz = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"  queue {q}: {len(rs):4d} launches, first start {1e-6 * s:.3f} ms, last end {1e-6 * e:.3f} ms, busy {1e-6 * bz:.3f} ms")
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as out:
        for r in win:
            s = int(r["Start_Timestamp"]) - t0
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            out.write(f"{1e-3 * s:9.1f} us  +{1e-3 * d:7.1f} us  q{r['Queue_Id']}  {r['Kernel_Name'][:110]}\n")

"""Print the main-path kernel sequence of the last steady-state step of a graph-mode bench trace (short names)."""
import csv, glob, re, sys
f = (glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv") + glob.glob(sys.argv[1] + "/*_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "fps_pruned_kernel" in r["Kernel_Name"]]
pairs = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if marks[i + 1] - marks[i] > 200]
a, b = pairs[-1]
side_names = ("fps_", "ball_query", "three_nn", "gather_points_kernel")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    m = re.search(r"(\w+Functor\w*|\w+_kernel_cuda|launch_\w+|masked_scale\w*|MaxNanFunctor|func_wrapper_t<\w+, \w+::(\w+)|sum_functor|MeanOps|WelfordOps|NormTwoOps)", n)
    base = n.split("(")[0][:60]
    return base + (" | " + m.group(0)[:50] if m and m.group(0) not in base else "")
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b]:
    if any(s in r["Kernel_Name"] for s in side_names): continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {short(r['Kernel_Name'])}")
    prev_end = max(prev_end, e)

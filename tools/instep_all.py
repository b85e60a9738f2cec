"""Every C entry point of one replayed step with its duration INSIDE the step (clock stamps captured into the graphs,
_lib.Stamps): launch order per stream, start offset, microseconds (bracket = two dispatch gaps subtracted), and the totals
per entry point.  The instrumented step is slower than the plain one (two one-thread kernels per call): read the durations,
not the sum.   python tools/instep_all.py [fp32]"""
import importlib
import os
import re
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ext = importlib.import_module("3dvlp_amd._lib")
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")

dev = torch.device("cuda:0")
skip = re.compile(r"bytes|slabs|nparam|splits|version|fp_contract|max_instances|param_floats|stamp|probe|hwprobe|blocks$")
names = [n for n, a in ext.SIGNATURES.items() if a and a[-1] is ext._vp and not skip.search(n)]
only = [a[5:].split(",") for a in sys.argv[1:] if a.startswith("only=")]   # only=copy_batch,adamw_flat: few stamps = the step's
if only:                                                                    # real timeline (every stamp pair costs ~3.5 us)
    names = [n for n in names if n[6:] in only[0]] + ["vlp3d_sa_fwd_gather"]
batch = gs.batch_to_device(synth.make_batch(0, 8, num_points=40000, lang_num_max=8), dev, feat_bf16=os.environ.get("VLP3D_FEAT_BF16", "1") != "0" and "fp32" not in sys.argv)
step = gs.GroundingStep(dev, epoch=50, sa_dtype=None if "fp32" in sys.argv else torch.bfloat16, use_graph=True, pipeline=True)
with ext.Stamps(names + ["vlp3d_probe_empty"], dev, capacity=4096) as st:
    def begin():
        st.log.clear()
        st.calibrate_pending = True
    step.on_capture = begin
    step.run(batch)
    samples = []
    for _ in range(10):
        step.run(batch)
        torch.cuda.synchronize()
        t = st.buf.cpu().tolist()
        samples.append(t)
    log = list(st.log)
med = [sorted(col)[len(col) // 2] for col in zip(*[[s[2 * k + 1] - s[2 * k] for k in range(len(log))] for s in samples[2:]])]
last = samples[-1]
gap = next(m for (n, _), m in zip(log, med) if n == "stamp_gap") * st.TICK_US
t0 = min(last[2 * k] for k in range(len(log)))
tot = defaultdict(lambda: [0.0, 0])
print(f"stamp gap {gap:.2f} us; {len(log)} stamped calls")
for k, ((n, a), m) in enumerate(zip(log, med)):
    us = m * st.TICK_US - 2 * gap
    print(f"{(last[2 * k] - t0) * st.TICK_US:9.1f} us  {us:7.1f} us  {n[6:]}  {a[:8]}")
    tot[n][0] += us
    tot[n][1] += 1
print("---- totals")
for n, (us, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"{us:9.1f} us {c:4d}x  {n}")
print(f"sum {sum(v[0] for v in tot.values()):.1f} us")

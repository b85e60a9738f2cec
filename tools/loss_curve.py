"""Loss over N steps on one repeated batch for a configuration (sanity of reduced-precision variants)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
dev = torch.device("cuda:0")
dt = {"bf16": torch.bfloat16, "fp32": None}[sys.argv[1]]
steps = int(sys.argv[2])
for seed in (0, 1):
    step = gs.GroundingStep(dev, sa_dtype=dt, use_graph=True, pipeline=True, seed=seed)
    batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
    ls = []
    for i in range(steps):
        ls.append(float(step.run(batch)))
    print(sys.argv[1], os.environ.get("VLP3D_LINEAR_BF16", "1"), "seed", seed, "first", round(ls[0], 3),
          "mean[40:60]", round(sum(ls[40:60]) / 20, 3), "mean[-20:]", round(sum(ls[-20:]) / 20, 3), flush=True)

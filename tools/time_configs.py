"""Step time of the other BASELINE configurations' per-GPU shapes on ONE MI355X (captured, pipelined, bf16): cfg3's
pre-training batch (32 scenes on 8 GPUs = 4 per GPU, and all 32 on one GPU), cfg4's caption head attached to the step
(GroundingStep(use_caption=True): 8 sentences x 32 tokens, 30 522 words), cfg5's 80 000-point joint QA + grounding step."""
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
dev = torch.device("cuda:0")
side = None
for name, B, npts, kw in (("cfg2  8 x 40k", 8, 40000, {}), ("cfg3  4 x 40k (32 scenes / 8 GPUs)", 4, 40000, {}),
                          ("cfg3 32 x 40k (one GPU)", 32, 40000, {}),
                          ("cfg4  8 x 40k, grounding + caption", 8, 40000, dict(use_caption=True)),
                          ("cfg4  4 x 40k, grounding + caption", 4, 40000, dict(use_caption=True)),
                          ("cfg5  4 x 80k, QA + grounding", 4, 80000, dict(use_answer=True, num_answers=512)),
                          ("cfg5  8 x 80k, QA + grounding", 8, 80000, dict(use_answer=True, num_answers=512))):
    batch = gs.batch_to_device(synth.make_batch(0, B, npts, 8, num_answers=kw.get("num_answers", 0),
                                                caption_tokens=32 if kw.get("use_caption") else 0), dev, feat_bf16=True)
    step = gs.GroundingStep(dev, epoch=50, sa_dtype=torch.bfloat16, use_graph=True, pipeline=True, side_stream=side, **kw)
    side = step._side
    for _ in range(12):
        step.run(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for _ in range(n):
        loss = step.run(batch)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(f"{name:38s} {ms:7.3f} ms/step  {B / ms * 1e3:8.1f} scenes/s   loss {float(loss):.3f}")
    del step, batch
    torch.cuda.empty_cache()

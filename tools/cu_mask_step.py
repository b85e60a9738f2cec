"""The cfg2 step with the MAIN stream CU-masked to 248 CUs (one CU per 32-bit mask word left out), everything else as in
bench.py: what do the dense kernels lose?   python tools/cu_mask_step.py [plain]"""
import ctypes
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
gs = importlib.import_module("3dvlp_amd.grounding_step")
synth = importlib.import_module("3dvlp_amd.synth")
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr) == 0
    return torch.cuda.ExternalStream(s.value, device=dev)


main = torch.cuda.Stream() if "plain" in sys.argv else masked_stream([0xFFFFFFFE] * 8)
batch = gs.batch_to_device(synth.make_batch(0, 8, 40000, 8), dev)
with torch.cuda.stream(main):
    step = gs.GroundingStep(dev, epoch=50, sa_dtype=torch.bfloat16, use_graph=True, pipeline=True)
    for _ in range(20):
        step.run(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        step.run(batch)
    torch.cuda.synchronize()
    print("main stream", "plain" if "plain" in sys.argv else "248 CUs", f"{1e3 * (time.perf_counter() - t0) / 100:.3f} ms/step")

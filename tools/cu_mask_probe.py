"""Does a CU-masked HIP stream keep the SA1 FPS (one workgroup per scene, 8 CUs) out of the dense kernels' way?
hipExtStreamCreateWithCUMask: stream A = 8 CUs (bits 0..7), stream B = the other 248; FPS alone, FPS beside a dense load on
plain streams, FPS beside the same load with both streams masked."""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
synth = importlib.import_module("3dvlp_amd.synth")
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    arr = (ctypes.c_uint32 * len(words))(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


ncu = torch.cuda.get_device_properties(dev).multi_processor_count
nw = (ncu + 31) // 32
few = [0] * nw
few[0] = 0xFF
rest = [0xFFFFFFFF] * nw
rest[0] = 0xFFFFFF00
print("CUs", ncu, "mask words", nw)
xyz = torch.from_numpy(synth.make_batch(0, 8, num_points=40000, lang_num_max=1)["point_clouds"][..., :3].copy()).to(dev)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)


def run(fps_stream, load_stream, with_load):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if with_load:
        with torch.cuda.stream(load_stream):
            for _ in range(12):
                c = a @ b
    with torch.cuda.stream(fps_stream):
        e0.record()
        pu.furthest_point_sample(xyz, 2048)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


if len(sys.argv) > 1 and sys.argv[1] == "spread":  # one CU per 32-bit word instead of eight neighbours
    few = [1] * nw
    rest = [0xFFFFFFFE] * nw
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
m1, m2 = masked_stream(few), masked_stream(rest)
for name, fs, ls, load in (("alone, plain stream", s1, s2, False), ("alone, 8-CU stream", m1, m2, False),
                           ("beside GEMMs, plain streams", s1, s2, True), ("beside GEMMs, masked streams", m1, m2, True)):
    ts = [run(fs, ls, load) for _ in range(4)]
    print(f"{name:32s} FPS {min(ts[1:]):.3f} ms")
# how much does the dense load lose on 248 CUs?
for name, ls in (("GEMMs, plain stream", s2), ("GEMMs, 248-CU stream", m2)):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(ls):
        e0.record()
        for _ in range(12):
            c = a @ b
        e1.record()
    torch.cuda.synchronize()
    print(f"{name:32s} {e0.elapsed_time(e1):.3f} ms")

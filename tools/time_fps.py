"""FPS kernel timing at several N (which storage classes are active: registers / LDS / global tail)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
synth = importlib.import_module("3dvlp_amd.synth")
xyz_full = torch.from_numpy(np.stack([synth.make_scene(1000 + i, 45000)["xyz"] for i in range(8)])).cuda()
for N, m in [(int(a.split(",")[0]), int(a.split(",")[1])) for a in sys.argv[1:]] or ((16384, 2048), (24576, 2048), (33792, 2048), (36864, 2048), (40000, 2048), (45000, 2048), (2048, 1024), (1024, 512), (512, 256)):
    x = xyz_full[:, :N].contiguous()
    for _ in range(2): pu.furthest_point_sample(x, m)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): pu.furthest_point_sample(x, m)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"N={N:6d} m={m:5d}: {ms:7.3f} ms  {1e3*ms/(m-1):6.3f} us/iter  {ms*1e-3*2.4e9/(m-1):7.0f} cyc/iter(2.4GHz)", flush=True)

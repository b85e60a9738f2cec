"""For a holder of the reference's built extension: which fp32 evaluation order does `pointnet2._ext` use?

    python tools/check_contract_vectors.py              # this repo's HIP library, all three modes (needs a MI355X)
    python tools/check_contract_vectors.py --ext        # `import pointnet2._ext as _ext` (the reference build, CUDA)

Runs the inputs of tests/golden/contract_vectors.npz (tests/golden/make_contract_vectors.py) through
furthest_point_sampling / ball_query / three_nn and reports, per case and per mode, how many index entries differ from the
file.  The mode with zero differences everywhere is the reference build's; select it here with
3dvlp_amd._lib.set_fp_contract(mode) or VLP3D_FP_CONTRACT=mode."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
V = np.load(os.path.join(ROOT, "tests", "golden", "contract_vectors.npz"))
names = sorted({k.split("/")[0] for k in V.files})
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run(ext, name):
    if name.startswith("fps"):
        return ext.furthest_point_sampling(dev(V[name + "/xyz"]), int(V[name + "/npoint"])).cpu().numpy()
    if name.startswith("bq"):
        return ext.ball_query(dev(V[name + "/new_xyz"]), dev(V[name + "/xyz"]), float(V[name + "/radius"]),
                              int(V[name + "/nsample"])).cpu().numpy()
    return ext.three_nn(dev(V[name + "/unknown"]), dev(V[name + "/known"]))[1].cpu().numpy()


if "--ext" in sys.argv:
    import pointnet2._ext as ext   # the reference's pybind module (lib/pointnet2/_ext_src/src/bindings.cpp:12-23)
    for name in names:
        got = run(ext, name)
        print(name, {mode: int((got != V[f"{name}/idx_mode{mode}"]).sum()) for mode in (0, 1, 2)})
else:
    ext = importlib.import_module("3dvlp_amd._lib")
    for mode in (0, 1, 2):
        ext.set_fp_contract(mode)
        bad = {name: int((run(ext, name) != V[f"{name}/idx_mode{mode}"]).sum()) for name in names}
        print("HIP library, mode", mode, "differences:", bad)
    ext.set_fp_contract(None)

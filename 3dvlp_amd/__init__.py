"""3dvlp_amd — MI355X (gfx950) implementation of the 3DVLP point-cloud + language grounding hot path.

The directory name is not a Python identifier; import it with
``importlib.import_module("3dvlp_amd")`` — the package also registers itself under the alias
``vlp3d_amd`` so that ``import vlp3d_amd.pointnet2_utils`` works afterwards.

Product code only: every op here runs through the hand-written HIP kernels of
``csrc/libvlp3d_hip.so`` (C ABI: include/vlp3d.h).  There is no CPU fallback — like the reference's
``pointnet2._ext`` the ops raise on CPU tensors, and importing an op module fails loudly when the
library is missing.
"""
import sys as _sys

_sys.modules.setdefault("vlp3d_amd", _sys.modules[__name__])

__all__ = ["_lib", "pointnet2_utils", "pointnet2_modules", "pytorch_utils", "nn_distance"]

"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/vlp3d.h declares; the Python mirror refuses CPU tensors like the reference does."""
import ctypes
import importlib
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "vlp3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long long|const char \*)\s*(vlp3d_\w+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    build = importlib.import_module("3dvlp_amd.build")
    return build.build(verbose=False)


def test_header_declares_the_nine_ext_ops():
    names = _declared()
    for op in ("furthest_point_sampling", "gather_points", "gather_points_grad", "ball_query", "group_points",
               "group_points_grad", "three_nn", "three_interpolate", "three_interpolate_grad"):
        assert "vlp3d_" + op in names


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built)
    for name in _declared():
        assert hasattr(lib, name), name
    lib.vlp3d_abi_version.restype = ctypes.c_int
    assert lib.vlp3d_abi_version() >= 1
    assert lib.vlp3d_fp_contract() in (0, 1, 2)


def test_binding_table_matches_header(built):
    _lib = importlib.import_module("3dvlp_amd._lib")
    assert sorted(_lib.SIGNATURES) == _declared()
    _lib.load()


def test_invalid_arguments_are_rejected_without_a_gpu(built):
    lib = ctypes.CDLL(built)
    # null pointers / non-positive extents -> VLP3D_EINVAL before any HIP call
    assert lib.vlp3d_furthest_point_sampling(None, 1, 8, 4, None, None, None) == -22
    assert lib.vlp3d_ball_query(None, None, 1, 8, 4, ctypes.c_float(0.1), 4, None, None) == -22
    assert lib.vlp3d_three_nn(None, None, 0, 1, 1, None, None, None) == -22


def test_cpu_tensors_raise_like_the_reference(built):
    pu = importlib.import_module("3dvlp_amd.pointnet2_utils")
    with pytest.raises(RuntimeError, match="CPU not supported"):
        pu.furthest_point_sample(torch.zeros(1, 8, 3), 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        pu.gather_operation(torch.zeros(1, 8, 3).transpose(1, 2), torch.zeros(1, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="int tensor"):
        pu.gather_operation(torch.zeros(1, 3, 8), torch.zeros(1, 2, dtype=torch.int64))


def test_state_dict_keys_match_reference_layout(golden, built):
    pt = importlib.import_module("3dvlp_amd.pytorch_utils")
    g = golden("shared_mlp")
    m = pt.SharedMLP([135, 64, 64, 128], bn=True)
    assert sorted(m.state_dict().keys()) == sorted(g["meta/keys"].tolist())

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


def test_geometry_variant_libraries_export_the_geometry_entry_points(built):
    """libvlp3d_geom_c0.so / _c2.so (build.py: the geometry sources under the two other fp32 evaluation orders) export every
    entry point _lib routes through them and report their mode; set_fp_contract switches and restores."""
    build = importlib.import_module("3dvlp_amd.build")
    _lib = importlib.import_module("3dvlp_amd._lib")
    for mode in build.GEOM_MODES:
        lib = ctypes.CDLL(build.geom_lib(mode))
        for name in _lib.GEOM_ENTRY_POINTS:
            assert hasattr(lib, name), (mode, name)
        assert lib.vlp3d_fp_contract() == mode
        assert lib.vlp3d_three_nn(None, None, 0, 1, 1, None, None, None) == -22
    assert _lib.fp_contract() == 1
    prev = _lib.set_fp_contract(2)
    try:
        assert _lib.fp_contract() == 2 and _lib._geom() is not _lib.load()
        with pytest.raises(ValueError):
            _lib.set_fp_contract(3)
    finally:
        _lib.set_fp_contract(prev)
    assert _lib.fp_contract() == 1 and _lib._geom() is _lib.load()


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


def test_descriptor_structs_have_the_header_layout(tmp_path):
    """The batched / chained entry points take arrays of plain C structs (include/vlp3d.h); the ctypes mirrors in _lib.py must
    agree with what a C compiler makes of the header: size and every field offset, from a probe compiled with gcc."""
    import subprocess
    _lib = importlib.import_module("3dvlp_amd._lib")
    pairs = {"vlp3d_copy_desc": _lib.CopyDesc, "vlp3d_transpose_desc": _lib.TransposeDesc, "vlp3d_chain_stage": _lib.ChainStage,
             "vlp3d_chain_bwd_point": _lib.ChainBwdPoint, "vlp3d_chain_bwd_gemm": _lib.ChainBwdGemm,
             "vlp3d_linear_wgrad_job": _lib.LinearWgradJob, "vlp3d_rows_wgrad_job": _lib.RowsWgradJob}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vlp3d.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for field, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, field, cname, field))
    lines += ['  return 0;', '}']
    src, exe = tmp_path / "probe.c", tmp_path / "probe"
    src.write_text("\n".join(lines))
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        cname, field, val = ln.split()
        got[(cname, field)] = int(val)
    for cname, cls in pairs.items():
        assert got[(cname, "size")] == ctypes.sizeof(cls), cname
        for field, _ in cls._fields_:
            assert got[(cname, field)] == getattr(cls, field).offset, (cname, field)


def test_row_chain_declines_what_its_kernels_do_not_cover():
    """Host logic of row_chain.supported: CPU tensors, the exact-fp32 configuration, row counts that fill no 64-row pair of
    tiles and widths outside the 128-column weight blocks go to the layer modules' own launches."""
    rc = importlib.import_module("3dvlp_amd.row_chain")
    ml = importlib.import_module("3dvlp_amd.mfma_linear")
    w = torch.zeros(128, 128)
    st = [rc.linear(w, None)]
    with ml.bf16_mma(True):
        assert not rc.supported(torch.zeros(64, 128), st)            # CPU
    assert not rc.supported(torch.zeros(64, 128), st)                # exact-fp32 configuration


def test_bf16_row_operand_combinations_are_checked_on_the_host():
    """vlp3d_sdpa_fwd_io / _bwd_io are built for io = 0, 5, 7 (include/vlp3d.h); the binding refuses every other mix of fp32 /
    bf16 operands before anything is launched, and the shapes the LDS kernels cover are a host-side rule (no GPU needed)."""
    import torch
    _lib = importlib.import_module("3dvlp_amd._lib")
    fa = importlib.import_module("3dvlp_amd.fused_attention")
    f, h = torch.zeros(1, 4, 128), torch.zeros(1, 4, 128, dtype=torch.bfloat16)
    assert _lib._sdpa_io(f, f, f, False) == 0
    assert _lib._sdpa_io(h, f, f, True) == 5
    assert _lib._sdpa_io(h, h, h, True) == 7
    for q, k, v, o in ((f, h, h, False), (h, f, f, False), (h, h, f, True), (f, f, f, True)):
        with pytest.raises(RuntimeError):
            _lib._sdpa_io(q, k, v, o)
    assert fa.rows_supported(256, 256, 256) and fa.rows_supported(256, 49, 64)
    assert not fa.rows_supported(256, 300, 256) and not fa.rows_supported(600, 49, 256) and not fa.rows_supported(256, 49, 32)

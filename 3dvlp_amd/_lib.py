"""ctypes binding of libvlp3d_hip.so (C ABI declared in include/vlp3d.h).

This is the Python side of the drop-in boundary: it plays the role of the reference's pybind
module ``pointnet2._ext`` (lib/pointnet2/_ext_src/src/bindings.cpp:11-24) — same nine functions,
same argument checks and messages (include/utils.h:10-30) — but over a plain C ABI, with outputs
allocated here by torch (the library allocates nothing) and work enqueued on torch's current
HIP stream.  No CPU path exists: CPU tensors raise like the reference ("CPU not supported").
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libvlp3d_hip.so")

_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float

# name -> argtypes; every function returns int (0 ok, -22 EINVAL, >0 hipError_t)
SIGNATURES = {
    "vlp3d_abi_version": [],
    "vlp3d_fp_contract": [],
    "vlp3d_furthest_point_sampling": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "vlp3d_fps_prefix_check": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "vlp3d_furthest_point_sampling_cond": [_vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "vlp3d_fps_workspace_bytes": [_i, _i],
    "vlp3d_furthest_point_sampling_pruned": [_vp, _i, _i, _i, _vp, ctypes.c_longlong, _vp, _vp],
    "vlp3d_gather_points": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "vlp3d_gather_points_grad": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "vlp3d_ball_query": [_vp, _vp, _i, _i, _i, _f, _i, _vp, _vp],
    "vlp3d_ball_query_sorted": [_vp, _vp, _i, _i, _i, _f, _i, _vp, ctypes.c_longlong, _vp, _vp],
    "vlp3d_ball_query_grid_workspace_bytes": [_i, _i],
    "vlp3d_ball_query_grid": [_vp, _vp, _i, _i, _i, _f, _i, _vp, ctypes.c_longlong, _vp, _vp],
    "vlp3d_group_points": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "vlp3d_group_points_grad": [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "vlp3d_three_nn": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "vlp3d_three_interpolate": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "vlp3d_three_interpolate_grad": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "vlp3d_nn_distance": [_vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp],
    "vlp3d_group_rows": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _i, _vp],
    "vlp3d_group_rows_grad": [_vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp],
    "vlp3d_sa_compact": [_vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "vlp3d_sa_inverse": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "vlp3d_sa_bwd_gather_csr": [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "vlp3d_sa_fwd_gather": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp],
    "vlp3d_sa_fwd_layer": [_vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp],
    "vlp3d_sa_pool": [_vp, ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "vlp3d_sa_pool_rows": [_vp, ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "vlp3d_sa_pool_grad": [_vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _i, _vp],
    "vlp3d_sa_bwd_layer": [_vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp],
    "vlp3d_sa_bwd_gather": [_vp, _vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _vp, _vp, _i, _vp],
    "vlp3d_sa_wgrad": [_vp, _vp, ctypes.c_longlong, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i,
                       _i, _f, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp],
    "vlp3d_slab_reduce_batch": [_vp, _i, _vp],
    "vlp3d_linear_wgrad_batch": [_vp, _i, _vp],
    "vlp3d_rows_wgrad_batch": [_vp, _i, _vp],
    "vlp3d_copy_batch": [_vp, _i, _vp],
    "vlp3d_smallk_fwd": [_vp, _i, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _vp],
    "vlp3d_smallk_bwd": [_vp, _vp, _i, ctypes.c_longlong, _i, _i, _vp, _vp],
    "vlp3d_rowdot_fwd": [_vp, _vp, _vp, ctypes.c_longlong, _i, _vp, _vp],
    "vlp3d_rowdot_bwd": [_vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _vp, _vp],
    "vlp3d_sa_stat_slabs": [ctypes.c_longlong],
    "vlp3d_sa_last_supported": [_i, _i],
    "vlp3d_sa_last_dgrad": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp],
    "vlp3d_sa_last_wgrad": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp],
    "vlp3d_sa_bn_fold": [_vp, _i, _vp, _vp, _vp, _vp, _i, ctypes.c_longlong, _f, _f, _i, _vp, _vp],
    "vlp3d_sa_bn_bwd_consts": [_vp, _vp, _vp, _i, _i, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp],
    "vlp3d_sa_pool_tstats_slabs": [ctypes.c_longlong],
    "vlp3d_sa_pool_tstats": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _vp],
    "vlp3d_sa_prep_weights": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "vlp3d_joint_loss_rows": [_i, _i, _i, _i],
    "vlp3d_joint_loss_fwd": [_vp] * 25 + [_i] * 8 + [_f] * 6 + [_i] + [_vp] * 6 + [_vp],
    "vlp3d_joint_loss_bwd": [_vp] * 25 + [_i] * 8 + [_f] * 6 + [_i] + [_vp] * 5 + [_vp] * 10 + [_vp],
    "vlp3d_joint_loss_report": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "vlp3d_contrast_fwd": [_vp] * 9 + [_i] * 4 + [_vp, _vp, _vp],
    "vlp3d_contrast_bwd": [_vp] * 9 + [_i] * 4 + [_vp] * 7 + [_vp],
    "vlp3d_add_norm_blocks": [ctypes.c_longlong],
    "vlp3d_add_norm_fwd": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _f, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp],
    "vlp3d_add_norm_bwd": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "vlp3d_add_norm_rep_fwd": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _f, _vp, _i, _f, _vp, _vp, _vp, _vp],
    "vlp3d_rep_sum2": [_vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp, _vp, _vp],
    "vlp3d_sum_norm_fwd": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _f, _vp, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "vlp3d_sum_norm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _f, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "vlp3d_act_dropout": [_vp, _vp, ctypes.c_longlong, _i, _f, _vp, _i, _vp, _vp, _vp],
    "vlp3d_box_decode_fwd": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "vlp3d_box_decode_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "vlp3d_linear_fwd": [_vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _i, _vp],
    "vlp3d_linear_dgrad": [_vp, _vp, ctypes.c_longlong, _i, _i, _vp, _vp, _i, _vp],
    "vlp3d_linear_wgrad": [_vp, _vp, ctypes.c_longlong, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "vlp3d_relation_bias_nparam": [],
    "vlp3d_relation_bias_fwd": [_vp, _vp, _i, _i, _vp, _vp],
    "vlp3d_relation_bias_bwd": [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _vp],
    "vlp3d_rows_slabs": [ctypes.c_longlong],
    "vlp3d_rows_fwd": [_vp, _i, ctypes.c_longlong, _i, _vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp],
    "vlp3d_rows_fwd_wt": [_vp, _i, ctypes.c_longlong, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _vp],
    "vlp3d_transpose_batch": [_vp, _i, _vp],
    "vlp3d_rows_dgrad": [_vp, _vp, _i, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _vp],
    "vlp3d_rows_wgrad": [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _vp],
    "vlp3d_rows_act": [_vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp],
    "vlp3d_rows_act_slabs": [ctypes.c_longlong],
    "vlp3d_rows_act_bwd": [_vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "vlp3d_fp_rows": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "vlp3d_fp_rows_grad": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "vlp3d_fp_rows_grad_csr": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "vlp3d_roi_split": [_vp, _i, ctypes.c_longlong, _i, _i, _f] + [_vp] * 8 + [_vp],
    "vlp3d_roi_split_bwd": [_vp] * 7 + [ctypes.c_longlong, _i, _i, _f, _vp, _i, _vp],
    "vlp3d_vote_epilogue": [_vp, _vp, _vp, _i, ctypes.c_longlong, _i, _vp, _vp, _vp, _vp],
    "vlp3d_vote_epilogue_bwd": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _vp, _vp, _i, _vp],
    "vlp3d_l2norm_rows": [_vp, ctypes.c_longlong, _i, _f, _vp, _vp, _vp],
    "vlp3d_l2norm_rows_bwd": [_vp, _vp, _vp, ctypes.c_longlong, _i, _f, _vp, _vp],
    "vlp3d_relation_inputs": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "vlp3d_relation_inputs_bf16": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "vlp3d_copy_paste_map": [_vp, _i, _i, _vp, _vp, _vp],
    "vlp3d_gather_rows": [_vp, _vp, ctypes.c_longlong, _i, _vp, _vp],
    "vlp3d_scatter_rows_add": [_vp, _vp, ctypes.c_longlong, _i, _vp, _vp],
    "vlp3d_adamw_flat": [_vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _f, _f, _f, _f, _f, _f, _f, _vp],
    "vlp3d_probe_read": [_vp, ctypes.c_longlong, _i, _vp, _vp],
    "vlp3d_probe_mfma_bf16": [_i, _i, _vp, _vp],
    "vlp3d_probe_fma_f32": [_i, _i, _vp, _vp],
    "vlp3d_fps_pruned_profile": [_vp, _i, _i, _i, _vp, ctypes.c_longlong, _vp, _vp, _vp],
    "vlp3d_fps_pruned_trace": [_vp, _i, _i, _i, _vp, ctypes.c_longlong, _vp, _vp, _vp, _i, _i, _vp],
    "vlp3d_stamp": [_vp, _vp],
    "vlp3d_probe_empty": [_i, _i, _vp, _vp],
    "vlp3d_gather_xyz": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "vlp3d_gather_xyz_grad": [_vp, _vp, _i, _i, _i, _vp, _vp],
    "vlp3d_three_nn_weights": [_vp, ctypes.c_longlong, _vp, _vp, _vp],
    "vlp3d_sa_bn_fold_shift": [_vp, _i, _vp, _vp, _vp, _vp, _i, ctypes.c_longlong, _f, _f, _i, _vp, _vp, _vp],
    "vlp3d_augment_param_floats": [],
    "vlp3d_augment_max_instances": [],
    "vlp3d_augment_points": [_vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp],
    "vlp3d_augment_votes": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "vlp3d_augment_boxes": [_vp, _i, _i, _vp, _vp, _vp],
    "vlp3d_bce_logits_blocks": [ctypes.c_longlong],
    "vlp3d_bce_logits_fwd": [_vp, _vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _vp, _vp],
    "vlp3d_bce_logits_bwd": [_vp, _vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _vp, _vp],
    "vlp3d_loss_tail_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp],
    "vlp3d_loss_tail_bwd": [_vp, _i, _i, _f, _f, _f, _vp, _vp],
    "vlp3d_cap_attn_fwd": [_vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp],
    "vlp3d_cap_attn_bwd": [_vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "vlp3d_vocab_ce_splits": [ctypes.c_longlong, _i],
    "vlp3d_vocab_ce_partial_bytes": [ctypes.c_longlong, _i],
    "vlp3d_vocab_ce_fwd": [_vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "vlp3d_vocab_ce_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _vp, _vp, _vp],
    "vlp3d_rows_chain": [_vp, ctypes.c_longlong, _vp, _i, _vp, _vp],
    "vlp3d_rows_chain_bwd_blocks": [ctypes.c_longlong],
    "vlp3d_rows_chain_bwd": [_vp, ctypes.c_longlong, _vp, _vp, _i, _vp, _vp],
    "vlp3d_sdpa_fwd": [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "vlp3d_sdpa_bwd": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                       _i, _i, _i, _i, _vp],
    "vlp3d_sdpa_fwd_io": [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "vlp3d_sdpa_bwd_io": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                          _i, _i, _i, _i, _i, _vp],
    "vlp3d_rows_chain_io": [_vp, _i, ctypes.c_longlong, _vp, _i, _vp, _vp],
    "vlp3d_linear_fwd_rows16": [_vp, _vp, _vp, ctypes.c_longlong, _i, _i, _vp, _vp],
}

_lib = None

# ---- fp32 evaluation order of the index-producing ops (include/vlp3d.h vlp3d_fp_contract) --------------
# The main library evaluates a*a + b*b + c*c as fma(c,c, fma(a,a, b*b)) (mode 1: what LLVM's NVPTX back end makes of the
# reference's expression under nvcc's default -fmad=true).  Whether the reference build really emits that form cannot be
# checked in this image, so the geometry ops also exist in the two other orders (libvlp3d_geom_c0.so: no contraction;
# libvlp3d_geom_c2.so: the left chain fma(c,c, fma(b,b, a*a))).  set_fp_contract(mode) — or VLP3D_FP_CONTRACT=0|1|2 in the
# environment at import — routes the nine `_ext` functions, the pruned / prefix FPS forms, the grid ball query and three_nn
# through the matching library; everything else (the fused layers) is independent of the mode.
GEOM_ENTRY_POINTS = (
    "vlp3d_abi_version", "vlp3d_fp_contract", "vlp3d_furthest_point_sampling", "vlp3d_fps_prefix_check",
    "vlp3d_furthest_point_sampling_cond", "vlp3d_fps_workspace_bytes", "vlp3d_furthest_point_sampling_pruned",
    "vlp3d_fps_pruned_profile", "vlp3d_fps_pruned_trace", "vlp3d_ball_query", "vlp3d_ball_query_grid_workspace_bytes",
    "vlp3d_ball_query_grid", "vlp3d_ball_query_sorted", "vlp3d_three_nn", "vlp3d_three_interpolate", "vlp3d_three_interpolate_grad", "vlp3d_gather_xyz",
    "vlp3d_gather_xyz_grad", "vlp3d_three_nn_weights", "vlp3d_gather_points", "vlp3d_gather_points_grad", "vlp3d_group_points",
    "vlp3d_group_points_grad")
_geom_libs = {}
_contract = None   # None = the main library's own mode


def _geom_path(mode):
    return os.path.join(_HERE, "csrc", "libvlp3d_geom_c%d.so" % mode)


def set_fp_contract(mode):
    """Select the fp32 evaluation order of the geometry ops: 0, 1 or 2 (None = the main library's, i.e. 1).  Returns the
    previous setting.  Fails loudly when the variant library has not been built (python 3dvlp_amd/build.py)."""
    global _contract
    prev = _contract
    if mode is not None:
        mode = int(mode)
        if mode not in (0, 1, 2):
            raise ValueError("fp contract mode must be 0, 1 or 2")
        if mode == load().vlp3d_fp_contract():
            mode = None
        elif mode not in _geom_libs:
            path = _geom_path(mode)
            if not os.path.exists(path):
                raise ImportError("%s not found — build it with `python 3dvlp_amd/build.py`" % path)
            lib = ctypes.CDLL(path)
            for name in GEOM_ENTRY_POINTS:
                fn = getattr(lib, name)
                fn.argtypes = SIGNATURES[name]
                fn.restype = ctypes.c_longlong if name.endswith(("_bytes", "_sums", "_rows")) else ctypes.c_int
            if lib.vlp3d_fp_contract() != mode:
                raise ImportError("%s reports fp contract mode %d" % (path, lib.vlp3d_fp_contract()))
            _geom_libs[mode] = lib
    _contract = mode
    return prev


def fp_contract():
    """The mode the geometry ops currently run in (0 / 1 / 2)."""
    return load().vlp3d_fp_contract() if _contract is None else _contract


def _geom():
    """The library that serves the geometry entry points under the current fp contract mode."""
    return load() if _contract is None else _geom_libs[_contract]


def load():
    """Load the library (once). Raises ImportError with build instructions when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libvlp3d_hip.so not found at %s — build it with `python 3dvlp_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_longlong if name.endswith(("_bytes", "_sums", "_rows")) else ctypes.c_int
        _lib = lib
        env = os.environ.get("VLP3D_FP_CONTRACT")
        if env not in (None, ""):
            set_fp_contract(int(env))
    return _lib


class Vlp3dError(RuntimeError):
    pass


def _check(status, name):
    if status != 0:
        what = "invalid argument" if status == -22 else "hipError_t %d" % status
        raise Vlp3dError("%s failed: %s" % (name, what))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _chk_float(t, name):
    if not t.is_contiguous():
        raise RuntimeError(name + " must be a contiguous tensor")
    if t.dtype != torch.float32:
        raise RuntimeError(name + " must be a float tensor")


def _chk_int(t, name):
    if not t.is_contiguous():
        raise RuntimeError(name + " must be a contiguous tensor")
    if t.dtype != torch.int32:
        raise RuntimeError(name + " must be an int tensor")


def _chk_dev(ref, *others):
    if not ref.is_cuda:
        raise RuntimeError("CPU not supported")
    for name, t in others:
        if not t.is_cuda or t.device != ref.device:
            raise RuntimeError(name + " must be a CUDA tensor")


# ---- the nine _ext functions (same names / argument order as bindings.cpp:12-23) ----
FPS_PRUNED_MIN_N = 8192   # below this the all-register dense kernel wins (no sort pre-pass)
FPS_PRUNED_MAX_N = 131072  # 64 slots per wave and lane-slot, two lane-slots above 65536 points


FPS_PREFIX_MAX_N = 65536


def furthest_point_sampling(points, nsamples, algorithm=None, prefix_hint=False, return_flag=False, return_workspace=False):
    """algorithm: None = pick by N, "dense" (csrc/fps.hip) or "pruned" (csrc/fps_pruned.hip) — identical output.

    prefix_hint=True says the caller EXPECTS `points` to be an earlier FPS's samples in sampling order (every backbone
    level after the first): two parallel kernels then try to prove that the result is 0..nsamples-1 and the sequential
    kernel runs only if the proof fails — the output is the same with or without the hint.  return_flag adds the
    device int (0 = proven) for tests.  return_workspace=True: -> (indices, workspace or None) — the pruned algorithm's
    workspace holds the cloud's spatial sort afterwards, which ball_query_sorted reads for the same cloud."""
    _chk_float(points, "points")
    _chk_dev(points)
    B, N, _ = points.shape
    out = torch.empty((B, nsamples), dtype=torch.int32, device=points.device)
    if prefix_hint and 1 <= nsamples <= N <= FPS_PREFIX_MAX_N:
        v = torch.empty((B, nsamples), dtype=torch.float32, device=points.device)
        flag = torch.empty((1,), dtype=torch.int32, device=points.device)
        tmp = torch.empty((B, N), dtype=torch.float32, device=points.device)
        with torch.cuda.device(points.device):
            _check(_geom().vlp3d_fps_prefix_check(_p(points), B, N, int(nsamples), _p(v), _p(flag), _stream()),
                   "fps_prefix_check")
            _check(_geom().vlp3d_furthest_point_sampling_cond(_p(points), B, N, int(nsamples), _p(tmp), _p(out),
                                                             _p(flag), _stream()), "furthest_point_sampling_cond")
        return (out, flag) if return_flag else out
    if algorithm is None:
        algorithm = "pruned" if FPS_PRUNED_MIN_N <= N <= FPS_PRUNED_MAX_N else "dense"
    ws = None
    with torch.cuda.device(points.device):
        if algorithm == "pruned":
            nbytes = int(_geom().vlp3d_fps_workspace_bytes(B, N))
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=points.device)
            _check(_geom().vlp3d_furthest_point_sampling_pruned(_p(points), B, N, int(nsamples), _p(ws), nbytes,
                                                               _p(out), _stream()), "furthest_point_sampling_pruned")
        else:
            tmp = torch.empty((B, N), dtype=torch.float32, device=points.device)
            _check(_geom().vlp3d_furthest_point_sampling(_p(points), B, N, int(nsamples), _p(tmp), _p(out), _stream()),
                   "furthest_point_sampling")
    return (out, ws) if return_workspace else out


def ball_query_sorted(new_xyz, xyz, radius, nsample, fps_workspace):
    """ball_query(new_xyz, xyz, radius, nsample) — identical output — in one launch on the spatial sort of `xyz` that
    furthest_point_sampling(xyz, ., "pruned", return_workspace=True) left in `fps_workspace` (csrc/ball_query_sorted.hip)."""
    _chk_float(new_xyz, "new_xyz")
    _chk_float(xyz, "xyz")
    _chk_dev(new_xyz, ("xyz", xyz), ("fps_workspace", fps_workspace))
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _check(_geom().vlp3d_ball_query_sorted(_p(new_xyz), _p(xyz), B, N, M, float(radius), int(nsample), _p(fps_workspace),
                                               fps_workspace.numel(), _p(idx), _stream()), "ball_query_sorted")
    return idx


def gather_points(points, idx):
    _chk_float(points, "points")
    _chk_int(idx, "idx")
    _chk_dev(points, ("idx", idx))
    B, C, N = points.shape
    M = idx.shape[1]
    out = torch.empty((B, C, M), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _check(_geom().vlp3d_gather_points(_p(points), _p(idx), B, C, N, M, _p(out), _stream()), "gather_points")
    return out


def gather_points_grad(grad_out, idx, n):
    _chk_float(grad_out, "grad_out")
    _chk_int(idx, "idx")
    _chk_dev(grad_out, ("idx", idx))
    B, C, M = grad_out.shape
    out = torch.empty((B, C, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _check(_geom().vlp3d_gather_points_grad(_p(grad_out), _p(idx), B, C, int(n), M, _p(out), _stream()),
               "gather_points_grad")
    return out


BALL_QUERY_GRID_MIN_N = 8192  # below this the all-pairs kernel (one pass over an L2-resident scene) wins


def ball_query(new_xyz, xyz, radius, nsample, algorithm=None):
    """algorithm: None = pick by N, "scan" (csrc/ball_query.hip) or "grid" (csrc/ball_query_grid.hip) — identical output."""
    _chk_float(new_xyz, "new_xyz")
    _chk_float(xyz, "xyz")
    _chk_dev(new_xyz, ("xyz", xyz))
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=xyz.device)
    if algorithm is None:
        algorithm = os.environ.get("VLP3D_BALL_QUERY") or ("grid" if N >= BALL_QUERY_GRID_MIN_N else "scan")
    with torch.cuda.device(xyz.device):
        if algorithm == "grid":
            nbytes = int(_geom().vlp3d_ball_query_grid_workspace_bytes(B, N))
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=xyz.device)
            _check(_geom().vlp3d_ball_query_grid(_p(new_xyz), _p(xyz), B, N, M, float(radius), int(nsample), _p(ws), nbytes,
                                                _p(idx), _stream()), "ball_query_grid")
        else:
            _check(_geom().vlp3d_ball_query(_p(new_xyz), _p(xyz), B, N, M, float(radius), int(nsample), _p(idx),
                                           _stream()), "ball_query")
    return idx


def group_points(points, idx):
    _chk_float(points, "points")
    _chk_int(idx, "idx")
    _chk_dev(points, ("idx", idx))
    B, C, N = points.shape
    _, M, S = idx.shape
    out = torch.empty((B, C, M, S), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _check(_geom().vlp3d_group_points(_p(points), _p(idx), B, C, N, M, S, _p(out), _stream()), "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    _chk_float(grad_out, "grad_out")
    _chk_int(idx, "idx")
    _chk_dev(grad_out, ("idx", idx))
    B, C, M, S = grad_out.shape
    out = torch.empty((B, C, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _check(_geom().vlp3d_group_points_grad(_p(grad_out), _p(idx), B, C, int(n), M, S, _p(out), _stream()),
               "group_points_grad")
    return out


def three_nn(unknowns, knows):
    _chk_float(unknowns, "unknowns")
    _chk_float(knows, "knows")
    _chk_dev(unknowns, ("knows", knows))
    B, n, _ = unknowns.shape
    m = knows.shape[1]
    dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknowns.device)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknowns.device)
    with torch.cuda.device(unknowns.device):
        _check(_geom().vlp3d_three_nn(_p(unknowns), _p(knows), B, n, m, _p(dist2), _p(idx), _stream()), "three_nn")
    return dist2, idx


def gather_xyz(xyz, idx):
    """xyz (B,N,3) f32, idx (B,M) i32 -> (B,M,3): the sampled coordinates, without the two transposes around gather_points."""
    _chk_float(xyz, "xyz")
    _chk_int(idx, "idx")
    _chk_dev(xyz, ("idx", idx))
    B, N, _ = xyz.shape
    M = idx.shape[1]
    out = torch.empty((B, M, 3), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _check(_geom().vlp3d_gather_xyz(_p(xyz), _p(idx), B, N, M, _p(out), _stream()), "gather_xyz")
    return out


def gather_xyz_grad(g, idx, N):
    _chk_float(g, "grad_out")
    _chk_int(idx, "idx")
    B, M, _ = g.shape
    out = torch.empty((B, int(N), 3), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _check(_geom().vlp3d_gather_xyz_grad(_p(g), _p(idx), B, int(N), M, _p(out), _stream()), "gather_xyz_grad")
    return out


def three_nn_weights(dist2):
    """dist2 (B,n,3) squared distances of three_nn -> inverse-distance weights (B,n,3) (pointnet2_modules.py:393-397)."""
    _chk_float(dist2, "dist2")
    _chk_dev(dist2)
    w = torch.empty_like(dist2)
    with torch.cuda.device(dist2.device):
        _check(_geom().vlp3d_three_nn_weights(_p(dist2), dist2.numel() // 3, _p(w), ctypes.c_void_p(0), _stream()), "three_nn_weights")
    return w


def three_interpolate(points, idx, weight):
    _chk_float(points, "points")
    _chk_int(idx, "idx")
    _chk_float(weight, "weight")
    _chk_dev(points, ("idx", idx), ("weight", weight))
    B, C, m = points.shape
    n = idx.shape[1]
    out = torch.empty((B, C, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _check(_geom().vlp3d_three_interpolate(_p(points), _p(idx), _p(weight), B, C, m, n, _p(out), _stream()),
               "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    _chk_float(grad_out, "grad_out")
    _chk_int(idx, "idx")
    _chk_float(weight, "weight")
    _chk_dev(grad_out, ("idx", idx), ("weight", weight))
    B, C, n = grad_out.shape
    out = torch.empty((B, C, m), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _check(_geom().vlp3d_three_interpolate_grad(_p(grad_out), _p(idx), _p(weight), B, C, n, int(m), _p(out),
                                                   _stream()), "three_interpolate_grad")
    return out


def sa_compact(idx, N):
    """Compact row map of a grouped MLP from the ball-query indices (csrc/sa_compact.hip): idx (B,M,S) i32, N = points per
    scene -> rowptr (B*M + 1) i32 with the number of distinct rows in the last entry, crow (B*M*S, 4) i32."""
    _chk_int(idx, "idx")
    B, M, S = idx.shape
    rowptr = torch.empty((B * M + 1,), dtype=torch.int32, device=idx.device)
    crow = torch.empty((B * M * S, 4), dtype=torch.int32, device=idx.device)
    with torch.cuda.device(idx.device):
        _check(load().vlp3d_sa_compact(_p(idx), B, int(N), M, S, _p(rowptr), _p(crow), _stream()), "sa_compact")
    return rowptr, crow


def sa_inverse(idx, N, cmap=None):
    """point -> rows map of a grouped MLP's gather (csrc/sa_compact.hip vlp3d_sa_inverse): idx (B,M,S) i32, cmap = (rowptr,
    crow) of sa_compact or None -> (inv_start (B*N+1) i32, inv_rows (B*M*S) i32)."""
    _chk_int(idx, "idx")
    B, M, S = idx.shape
    start = torch.empty((B * int(N) + 1,), dtype=torch.int32, device=idx.device)
    rows = torch.empty((B * M * S,), dtype=torch.int32, device=idx.device)
    cursor = torch.empty((B * int(N),), dtype=torch.int32, device=idx.device)
    rowptr, crow = cmap if cmap is not None else (None, None)
    with torch.cuda.device(idx.device):
        _check(load().vlp3d_sa_inverse(_p(idx), _opt(crow), _opt(rowptr), B, int(N), M, S, _p(start), _p(rows), _p(cursor),
                                       _stream()), "sa_inverse")
    return start, rows


def three_interpolate_grad_asshipped(grad_out, idx, weight, m):
    """What the reference's three_interpolate_grad EXECUTES (interpolate.cpp:77-104 calls the forward wrapper with
    m := n, n := m, points := grad_out): a forward blend of grad_out with a wrong batch stride, no scatter.  Only for
    like-for-like comparisons with a run of the reference; the default backward is the true adjoint."""
    _chk_float(grad_out, "grad_out")
    _chk_int(idx, "idx")
    _chk_float(weight, "weight")
    _chk_dev(grad_out, ("idx", idx), ("weight", weight))
    B, C, n = grad_out.shape
    m = int(m)
    if m > n:
        raise RuntimeError("three_interpolate_grad_asshipped: m > n would read idx / weight out of bounds "
                           "(the reference does, too)")
    out = torch.empty((B, C, m), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _check(_geom().vlp3d_three_interpolate(_p(grad_out), _p(idx), _p(weight), B, C, n, m, _p(out), _stream()),
               "three_interpolate (as-shipped gradient)")
    return out


# ---- fused ops ----
def nn_distance(pc1, pc2, mode, delta):
    _chk_float(pc1, "pc1")
    _chk_float(pc2, "pc2")
    _chk_dev(pc1, ("pc2", pc2))
    B, N, C = pc1.shape
    M = pc2.shape[1]
    if C != 3 or pc2.shape[2] != 3 or pc2.shape[0] != B:
        raise RuntimeError("nn_distance: expected (B,N,3) and (B,M,3)")
    dist1 = torch.empty((B, N), dtype=torch.float32, device=pc1.device)
    idx1 = torch.empty((B, N), dtype=torch.int64, device=pc1.device)
    dist2 = torch.empty((B, M), dtype=torch.float32, device=pc1.device)
    idx2 = torch.empty((B, M), dtype=torch.int64, device=pc1.device)
    with torch.cuda.device(pc1.device):
        _check(load().vlp3d_nn_distance(_p(pc1), _p(pc2), B, N, M, int(mode), float(delta), _p(dist1), _p(idx1),
                                        _p(dist2), _p(idx2), _stream()), "nn_distance")
    return dist1, idx1, dist2, idx2


def _opt(t):
    return ctypes.c_void_p(0) if t is None else ctypes.c_void_p(t.data_ptr())


def _row_stride(t, name):
    """Row stride (floats) of a (B, n, C) float32 CUDA tensor whose rows are contiguous and evenly spaced — a contiguous
    tensor or a column block of one (e.g. q / k / v slices of a merged projection)."""
    if not t.is_cuda:
        raise RuntimeError("CPU not supported")
    if t.dtype != torch.float32 or t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1) \
            or t.stride(1) % 4 or t.data_ptr() % 16:
        raise RuntimeError(name + " must be a float tensor with contiguous, evenly spaced, 16-byte aligned rows")
    return t.stride(1)


def _row_stride_ok(t):
    return (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1
            and t.stride(0) == t.shape[1] * t.stride(1) and t.stride(1) % 4 == 0 and t.data_ptr() % 16 == 0)


def _adjacent(a, b):
    """b is the column block right after a in the same row-strided buffer."""
    return (a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.stride() == b.stride()
            and a.shape[:2] == b.shape[:2] and b.storage_offset() == a.storage_offset() + a.shape[2])


def sdpa_fwd(q, k, v, H, bias, bias_mode, mask, bf16_mma=False):
    """q (B,nq,H*32), k/v (B,nk,H*32), each contiguous or a column block of a wider buffer
    -> (out (B,nq,H*32) contiguous, lse (B,H,nq))."""
    ldq, ldk, ldv = _row_stride(q, "q"), _row_stride(k, "k"), _row_stride(v, "v")
    _chk_dev(q, ("k", k), ("v", v))
    B, nq, HD = q.shape
    nk = k.shape[1]
    if bias is not None:
        _chk_float(bias, "attention_weights")
        if tuple(bias.shape) != (B, H, nq, nk):
            raise RuntimeError("attention_weights must be (b_s, h, nq, nk)")
    if mask is not None:
        _chk_float(mask, "attention_mask")
    out = torch.empty((B, nq, HD), dtype=torch.float32, device=q.device)
    lse = torch.empty((B, H, nq), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        _check(load().vlp3d_sdpa_fwd(_p(q), _p(k), _p(v), _opt(bias), int(bias_mode), _opt(mask), B, H, nq, nk,
                                     HD // H, _p(out), _p(lse), int(bool(bf16_mma)), ldq, ldk, ldv, _stream()),
               "sdpa_fwd")
    return out, lse


def sdpa_bwd(q, k, v, H, bias, bias_mode, mask, out, lse, dout, need_dbias, bf16_mma=False):
    """Gradients with the layout of their inputs: when q, k, v (or k, v) are adjacent column blocks of one buffer, dq, dk,
    dv (dk, dv) are column blocks of ONE gradient buffer, i.e. already the dY of the merged projection."""
    ldq, ldk, ldv = _row_stride(q, "q"), _row_stride(k, "k"), _row_stride(v, "v")
    B, nq, HD = q.shape
    nk = k.shape[1]
    _chk_float(dout, "dout")
    new = lambda n, c: torch.empty((B, n, c), dtype=torch.float32, device=q.device)
    if _adjacent(q, k) and _adjacent(k, v) and ldq == 3 * HD:
        g = new(nq, 3 * HD)
        dq, dk, dv = g[..., :HD], g[..., HD:2 * HD], g[..., 2 * HD:]
    elif _adjacent(k, v) and ldk == 2 * HD:
        g = new(nk, 2 * HD)
        dq, dk, dv = new(nq, HD), g[..., :HD], g[..., HD:]
        ldq = HD
    else:
        dq, dk, dv = new(nq, HD), new(nk, HD), new(nk, HD)
        ldq = ldk = ldv = HD
    # forward operands keep their own strides; the gradient strides follow the buffers allocated above
    gq, gk, gv = dq.stride(1), dk.stride(1), dv.stride(1)
    if (gq, gk, gv) != (_row_stride(q, "q"), _row_stride(k, "k"), _row_stride(v, "v")):
        # the kernels use one stride per operand for input and gradient: fall back to contiguous copies of the inputs
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        dq, dk, dv = new(nq, HD), new(nk, HD), new(nk, HD)
        gq = gk = gv = HD
    dbias = torch.empty((B, H, nq, nk), dtype=torch.float32, device=q.device) if need_dbias else None
    delta = torch.empty((B, H, nq), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        _check(load().vlp3d_sdpa_bwd(_p(q), _p(k), _p(v), _opt(bias), int(bias_mode), _opt(mask), _p(out), _p(lse),
                                     _p(dout), B, H, nq, nk, HD // H, _p(dq), _p(dk), _p(dv), _opt(dbias),
                                     _p(delta), int(bool(bf16_mma)), gq, gk, gv, _stream()), "sdpa_bwd")
    return dq, dk, dv, dbias


def _row_stride_bf(t, name):
    if not (t.is_cuda and t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1)):
        raise RuntimeError("%s must be (B, n, H*32) rows with one row stride" % name)
    if t.dtype == torch.bfloat16:
        if t.stride(1) % 8 or t.data_ptr() % 16:
            raise RuntimeError("%s: bf16 rows need 16-byte aligned rows" % name)
    elif t.dtype != torch.float32 or t.stride(1) % 4 or t.data_ptr() % 16:
        raise RuntimeError("%s must be fp32 or bf16 rows, 16-byte aligned" % name)
    return t.stride(1)


def _sdpa_io(q, k, v, out_bf16):
    io = int(q.dtype == torch.bfloat16) | (int(k.dtype == torch.bfloat16) << 1) | (int(bool(out_bf16)) << 2)
    if k.dtype != v.dtype or io not in (0, 5, 7):
        raise RuntimeError("sdpa rows: built combinations are fp32 everywhere, bf16 q / out with fp32 k / v, all bf16")
    return io


def sdpa_fwd_rows(q, k, v, H, mask, out_bf16):
    """The bf16-MFMA core on operands that are bf16 rows already (vlp3d_sdpa_fwd_io): q (B,nq,H*32) fp32 or bf16, k / v
    (B,nk,H*32) fp32 or bf16 (column blocks of a merged buffer allowed) -> (out (B,nq,H*32) bf16 if out_bf16 else fp32, lse)."""
    ldq, ldk, ldv = _row_stride_bf(q, "q"), _row_stride_bf(k, "k"), _row_stride_bf(v, "v")
    _chk_dev(q, ("k", k), ("v", v))
    io = _sdpa_io(q, k, v, out_bf16)
    B, nq, HD = q.shape
    nk = k.shape[1]
    if mask is not None:
        _chk_float(mask, "attention_mask")
    out = torch.empty((B, nq, HD), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=q.device)
    lse = torch.empty((B, H, nq), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        _check(load().vlp3d_sdpa_fwd_io(_p(q), _p(k), _p(v), None, 0, _opt(mask), B, H, nq, nk, HD // H, _p(out), _p(lse), 1,
                                        ldq, ldk, ldv, io, _stream()), "sdpa_fwd_io")
    return out, lse


def sdpa_bwd_rows(q, k, v, H, mask, out, lse, dout):
    """Backward of sdpa_fwd_rows: fp32 dq / dk / dv; k, v (q, k, v) adjacent column blocks of one buffer give ONE merged
    gradient buffer, as sdpa_bwd."""
    ldq, ldk, ldv = _row_stride_bf(q, "q"), _row_stride_bf(k, "k"), _row_stride_bf(v, "v")
    io = _sdpa_io(q, k, v, out.dtype == torch.bfloat16)
    B, nq, HD = q.shape
    nk = k.shape[1]
    _chk_float(dout, "dout")
    new = lambda n, c: torch.empty((B, n, c), dtype=torch.float32, device=q.device)
    if _adjacent(q, k) and _adjacent(k, v) and ldq == 3 * HD:
        g = new(nq, 3 * HD)
        dq, dk, dv = g[..., :HD], g[..., HD:2 * HD], g[..., 2 * HD:]
    elif _adjacent(k, v) and ldk == 2 * HD and ldq == HD:
        g = new(nk, 2 * HD)
        dq, dk, dv = new(nq, HD), g[..., :HD], g[..., HD:]
    else:
        raise RuntimeError("sdpa_bwd_rows: q | k | v merged, or q contiguous with k | v merged")
    delta = torch.empty((B, H, nq), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        _check(load().vlp3d_sdpa_bwd_io(_p(q), _p(k), _p(v), None, 0, _opt(mask), _p(out), _p(lse), _p(dout), B, H, nq, nk,
                                        HD // H, _p(dq), _p(dk), _p(dv), None, _p(delta), 1, ldq, ldk, ldv, io, _stream()),
               "sdpa_bwd_io")
    return dq, dk, dv


def linear_fwd_rows16(x2, w, bias):
    """x2 (R, K) fp32 -> (R, N) bf16 rows = x2 w^T + bias with bf16 MFMA operands (vlp3d_linear_fwd_rows16)."""
    _chk_float(x2, "x")
    _chk_float(w, "weight")
    R, K = x2.shape
    N = w.shape[0]
    y = torch.empty((R, N), dtype=torch.bfloat16, device=x2.device)
    with torch.cuda.device(x2.device):
        _check(load().vlp3d_linear_fwd_rows16(_p(x2), _p(w), _opt(bias), R, K, N, _p(y), _stream()), "linear_fwd_rows16")
    return y


def group_rows(xyz, new_xyz, idx, feat_pm, radius, out_dtype):
    """-> (B*M*S, C+4) rows [features | (xyz[idx]-new_xyz)/radius | 0] in out_dtype (float32 / bfloat16)."""
    _chk_float(xyz, "xyz")
    _chk_float(new_xyz, "new_xyz")
    _chk_int(idx, "idx")
    _chk_float(feat_pm, "features")
    _chk_dev(xyz, ("new_xyz", new_xyz), ("idx", idx), ("features", feat_pm))
    B, N, _ = xyz.shape
    _, M, S = idx.shape
    C = feat_pm.shape[2]
    if tuple(feat_pm.shape) != (B, N, C) or C % 4:
        raise RuntimeError("group_rows: features must be point-major (B,N,C) with C % 4 == 0")
    out = torch.empty((B * M * S, C + 4), dtype=out_dtype, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _check(load().vlp3d_group_rows(_p(xyz), _p(new_xyz), _p(idx), _p(feat_pm), B, N, M, S, C, float(radius),
                                       _p(out), int(out_dtype == torch.bfloat16), _stream()), "group_rows")
    return out


def group_rows_grad(dout, idx, B, N, C, radius, need_feat, need_xyz, need_new_xyz):
    _, M, S = idx.shape
    if not dout.is_contiguous() or dout.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError("group_rows_grad: dout must be a contiguous float/bfloat16 tensor")
    dev = dout.device
    dfeat = torch.empty((B, N, C), dtype=torch.float32, device=dev) if need_feat else None
    dxyz = torch.empty((B, N, 3), dtype=torch.float32, device=dev) if need_xyz else None
    dnew = torch.empty((B, M, 3), dtype=torch.float32, device=dev) if need_new_xyz else None
    with torch.cuda.device(dev):
        _check(load().vlp3d_group_rows_grad(_p(dout), int(dout.dtype == torch.bfloat16), _p(idx), B, N, M, S, C,
                                            float(radius), _opt(dfeat), _opt(dxyz), _opt(dnew), _stream()),
               "group_rows_grad")
    return dfeat, dxyz, dnew


class SlabReduceDesc(ctypes.Structure):
    """include/vlp3d.h: vlp3d_slab_reduce_desc."""
    _fields_ = [("partials", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("dbias", ctypes.c_void_p), ("nblk", _i),
                ("n_mat", _i), ("n_bias", _i), ("K", _i), ("ldo", _i), ("ncol_out", _i), ("rot", _i)]


class LinearWgradJob(ctypes.Structure):
    """include/vlp3d.h: vlp3d_linear_wgrad_job."""
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("partials", ctypes.c_void_p), ("R", ctypes.c_longlong),
                ("K", _i), ("N", _i), ("max_blocks", _i), ("with_bias", _i), ("x_bf16", _i)]


class RowsWgradJob(ctypes.Structure):
    """include/vlp3d.h: vlp3d_rows_wgrad_job."""
    _fields_ = [("G", ctypes.c_void_p), ("Ypre", ctypes.c_void_p), ("ldg", _i), ("bn5", ctypes.c_void_p),
                ("X", ctypes.c_void_p), ("lda", _i), ("a_scale", ctypes.c_void_p), ("a_shift", ctypes.c_void_p),
                ("R", ctypes.c_longlong), ("K", _i), ("N", _i), ("partials", ctypes.c_void_p), ("max_blocks", _i),
                ("with_bias", _i), ("x_bf16", _i)]


class CopyDesc(ctypes.Structure):
    """include/vlp3d.h: vlp3d_copy_desc."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("bytes", ctypes.c_longlong)]


class TransposeDesc(ctypes.Structure):
    """include/vlp3d.h: vlp3d_transpose_desc."""
    _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                ("ld_dst", ctypes.c_int)]


def transpose_batch(dsts, srcs):
    """dst[i] (cols x ld) = src[i] (rows x cols)^T for contiguous fp32 CUDA matrices — one launch (csrc/glue.hip)."""
    if not dsts:
        return
    arr = (TransposeDesc * len(dsts))()
    for c, d, s_ in zip(arr, dsts, srcs):
        if s_.dim() != 2 or d.dim() != 2 or not (s_.is_contiguous() and d.is_contiguous() and s_.is_cuda and d.is_cuda) or \
                s_.dtype != torch.float32 or d.dtype != torch.float32 or d.shape[0] != s_.shape[1] or d.shape[1] < s_.shape[0]:
            raise RuntimeError("transpose_batch: contiguous fp32 CUDA matrices, dst (cols x >= rows)")
        c.src, c.dst, c.rows, c.cols, c.ld_dst = s_.data_ptr(), d.data_ptr(), s_.shape[0], s_.shape[1], d.shape[1]
    with torch.cuda.device(dsts[0].device):
        _check(load().vlp3d_transpose_batch(ctypes.cast(arr, ctypes.c_void_p), len(dsts), _stream()), "vlp3d_transpose_batch")


def copy_batch(dsts, srcs):
    """dst[i].copy_(src[i]) for same-shape, same-dtype contiguous CUDA tensors — one launch (csrc/glue.hip)."""
    if not dsts:
        return
    arr = (CopyDesc * len(dsts))()
    for c, d, s_ in zip(arr, dsts, srcs):
        if d.shape != s_.shape or d.dtype != s_.dtype or not (d.is_contiguous() and s_.is_contiguous()) or \
                not (d.is_cuda and s_.is_cuda):
            raise RuntimeError("copy_batch: same-shape, same-dtype contiguous CUDA tensors only")
        c.src, c.dst, c.bytes = s_.data_ptr(), d.data_ptr(), d.numel() * d.element_size()
    with torch.cuda.device(dsts[0].device):
        _check(load().vlp3d_copy_batch(ctypes.cast(arr, ctypes.c_void_p), len(dsts), _stream()), "vlp3d_copy_batch")


class ChainStage(ctypes.Structure):
    """include/vlp3d.h: vlp3d_chain_stage."""
    _fields_ = [("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("N", ctypes.c_int), ("K", ctypes.c_int),
                ("v_out", ctypes.c_void_p), ("act_kind", ctypes.c_int), ("act_p", ctypes.c_float), ("act_call", ctypes.c_int),
                ("h_out", ctypes.c_void_p), ("has_ln", ctypes.c_int), ("res", ctypes.c_void_p), ("gamma", ctypes.c_void_p),
                ("beta", ctypes.c_void_p), ("ln_p", ctypes.c_float), ("ln_call", ctypes.c_int), ("eps", ctypes.c_float),
                ("ln_out", ctypes.c_void_p), ("xhat", ctypes.c_void_p), ("rstd", ctypes.c_void_p), ("v_out_bf16", ctypes.c_int),
                ("h_out_bf16", ctypes.c_int)]


def _prod(shape):
    n = 1
    for d in shape:
        n *= int(d)
    return n


def rows_chain(X, stages, seed):
    """One launch of csrc/rows_chain.hip.  X (R, K0) fp32 contiguous CUDA; stages: list of dicts with the fields of
    vlp3d_chain_stage (tensors or None for the pointers; missing keys = NULL / 0, act_kind defaults to -1)."""
    if not (X.is_cuda and X.dtype in (torch.float32, torch.bfloat16) and X.is_contiguous() and X.dim() == 2):
        raise RuntimeError("rows_chain: X must be a contiguous fp32 (or bf16 rows) CUDA matrix")
    R = X.shape[0]
    arr = (ChainStage * len(stages))()
    ptrs = ("W", "bias", "v_out", "h_out", "res", "gamma", "beta", "ln_out", "xhat", "rstd")
    width = {"bias": lambda st: (st["N"],), "v_out": lambda st: (R, st["N"]), "h_out": lambda st: (R, st["N"]),
             "res": lambda st: (R, st["N"]), "gamma": lambda st: (st["N"],), "beta": lambda st: (st["N"],),
             "ln_out": lambda st: (R, st["N"]), "xhat": lambda st: (R, st["N"]), "rstd": lambda st: (R,),
             "W": lambda st: (st["N"], st["K"])}
    for c, st in zip(arr, stages):
        for name in ptrs:
            t = st.get(name)
            if t is not None:
                want = torch.bfloat16 if ((name == "v_out" and st.get("v_out_bf16")) or
                                          (name == "h_out" and st.get("h_out_bf16"))) else torch.float32
                if not (t.is_cuda and t.dtype == want and t.is_contiguous() and t.device == X.device) or \
                        t.numel() != _prod(width[name](st)):
                    raise RuntimeError("rows_chain: %s must be contiguous fp32 on X's device with %s elements"
                                       % (name, width[name](st)))
            setattr(c, name, None if t is None else t.data_ptr())
        c.N, c.K = int(st["N"]), int(st["K"])
        c.act_kind, c.act_p, c.act_call = int(st.get("act_kind", -1)), float(st.get("act_p", 0.0)), int(st.get("act_call", 0))
        c.has_ln, c.ln_p, c.ln_call = int(st.get("has_ln", 0)), float(st.get("ln_p", 0.0)), int(st.get("ln_call", 0))
        c.eps = float(st.get("eps", 1e-5))
        c.v_out_bf16 = int(bool(st.get("v_out_bf16", 0)))
        c.h_out_bf16 = int(bool(st.get("h_out_bf16", 0)))
    if stages[0]["K"] != X.shape[1]:
        raise RuntimeError("rows_chain: X has %d columns, stage 0 reads %d" % (X.shape[1], stages[0]["K"]))
    with torch.cuda.device(X.device):
        _check(load().vlp3d_rows_chain_io(X.data_ptr(), int(X.dtype == torch.bfloat16), R, ctypes.cast(arr, ctypes.c_void_p),
                                          len(stages), None if seed is None else seed.data_ptr(), _stream()),
               "vlp3d_rows_chain_io")


class ChainBwdPoint(ctypes.Structure):
    """include/vlp3d.h: vlp3d_chain_bwd_point."""
    _fields_ = [("base", ctypes.c_void_p), ("add_kept", ctypes.c_int), ("op", ctypes.c_int), ("aux", ctypes.c_void_p),
                ("rstd", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("p", ctypes.c_float), ("call", ctypes.c_int),
                ("act_kind", ctypes.c_int), ("g_out", ctypes.c_void_p), ("dres_out", ctypes.c_void_p), ("keep", ctypes.c_int),
                ("part", ctypes.c_void_p), ("aux_bf16", ctypes.c_int)]


class ChainBwdGemm(ctypes.Structure):
    """include/vlp3d.h: vlp3d_chain_bwd_gemm."""
    _fields_ = [("Wt", ctypes.c_void_p), ("N", ctypes.c_int), ("K", ctypes.c_int)]


def rows_chain_bwd_blocks(R):
    return int(load().vlp3d_rows_chain_bwd_blocks(int(R)))


def rows_chain_bwd(G, points, gemms, seed):
    """One launch of the chain backward (csrc/rows_chain.hip).  G (R, gemms[0].K) fp32 contiguous; points: len(gemms) + 1 dicts
    with the fields of vlp3d_chain_bwd_point, gemms: dicts Wt (N, K) / N / K (include/vlp3d.h)."""
    R = G.shape[0]
    if not (G.is_cuda and G.dtype == torch.float32 and G.is_contiguous() and G.dim() == 2) or len(points) != len(gemms) + 1:
        raise RuntimeError("rows_chain_bwd: G must be a contiguous fp32 CUDA matrix, one more point than products")
    widths = [gemms[0]["K"]] + [g["N"] for g in gemms]
    if G.shape[1] != widths[0]:
        raise RuntimeError("rows_chain_bwd: G has %d columns, product 0 reduces over %d" % (G.shape[1], widths[0]))
    pa = (ChainBwdPoint * len(points))()
    ga = (ChainBwdGemm * len(gemms))()
    nblk = rows_chain_bwd_blocks(R)
    for c, P, N in zip(pa, points, widths):
        sizes = {"base": R * N, "aux": R * N, "rstd": R, "gamma": N, "g_out": R * N, "dres_out": R * N, "part": nblk * 2 * N}
        for name, n in sizes.items():
            t = P.get(name)
            want = torch.bfloat16 if (name == "aux" and P.get("aux_bf16")) else torch.float32
            if t is not None and (not (t.is_cuda and t.dtype == want and t.is_contiguous() and t.device == G.device)
                                  or t.numel() != n):
                raise RuntimeError("rows_chain_bwd: %s must be contiguous %s on G's device with %d elements" % (name, want, n))
            setattr(c, name, None if t is None else t.data_ptr())
        c.aux_bf16 = int(bool(P.get("aux_bf16", 0)))
        c.add_kept, c.op, c.keep = int(P.get("add_kept", 0)), int(P.get("op", 0)), int(P.get("keep", 0))
        c.p, c.call, c.act_kind = float(P.get("p", 0.0)), int(P.get("call", 0)), int(P.get("act_kind", 0))
    for c, g in zip(ga, gemms):
        t = g["Wt"]
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device == G.device) or \
                tuple(t.shape) != (g["N"], g["K"]):
            raise RuntimeError("rows_chain_bwd: Wt must be a contiguous fp32 (N, K) matrix on G's device")
        c.Wt, c.N, c.K = t.data_ptr(), int(g["N"]), int(g["K"])
    with torch.cuda.device(G.device):
        _check(load().vlp3d_rows_chain_bwd(G.data_ptr(), R, ctypes.cast(pa, ctypes.c_void_p), ctypes.cast(ga, ctypes.c_void_p),
                                           len(gemms), None if seed is None else seed.data_ptr(), _stream()),
               "vlp3d_rows_chain_bwd")


def wgrad_slabs(R, max_blocks):
    """Number of slabs a weight-gradient launch over R rows writes (the formula of csrc/sa_mlp.hip)."""
    tiles = R // 32
    tpb = max(1, (tiles + max_blocks - 1) // max_blocks)
    return (tiles + tpb - 1) // tpb


class SlabReduceQueue:
    """Weight-gradient launches issued while a queue is active (`deferred_slab_reduce`) leave their per-workgroup slabs
    behind; `flush` sums all of them with one launch per 40 entries (vlp3d_slab_reduce_batch).  The queue holds the
    tensors until then.  The gradients are valid only after the flush — the step driver flushes right after
    `loss.backward()`; code that reads `.grad` inside backward hooks must not use it.  A backward function must hand
    autograd a VIEW of `dst` (e.g. `dW.view(N, K)`), never `dst` itself: AccumulateGrad keeps a gradient tensor without
    copying only when nothing else references it, and the queue does — the copy would be taken before the sum ran."""

    FLUSH_BYTES = int(os.environ.get("VLP3D_SLAB_FLUSH_MB", 64)) << 20  # sum while the slabs are still cache-resident

    def __init__(self):
        self.items = []
        self.pending = 0
        self.added = 0
        self.wjobs = []    # queued weight-gradient launches of plain linear layers (vlp3d_linear_wgrad_batch) ...
        self.rjobs = []    # ... and of rows-stack layers (vlp3d_rows_wgrad_batch): (field dict incl. tensors, kept tensors) ...
        self.witems = []   # ... and their slab sums, which become due once the batches have run
        self.deferred = []  # (callable, kept tensors): other launches that feed only the optimiser (run first at the flush)

    def add(self, partials, nblk, dst, n_mat, K, ldo, dbias=None, n_bias=0, ncol_out=0, rot=0):
        self.items.append((partials, dst, dbias, int(nblk), int(n_mat), int(n_bias), int(K), int(ldo), int(ncol_out), int(rot)))
        self.added += 1
        self.pending += 4 * int(nblk) * (int(n_mat) + int(n_bias))
        if self.pending >= self.FLUSH_BYTES or len(self.items) >= 40:
            self.flush(everything=False)

    def add_linear_wgrad(self, dy, x, partials, R, K, N, max_blocks, with_bias, nblk, dst, dbias):
        """Queue the weight gradient of a plain linear layer itself (not only its slab sum): these launches are a few
        hundred workgroups each and feed nothing but the optimiser, so up to 48 of them run as ONE launch at the end of
        backward (csrc/sa_mlp.hip linear_wgrad_batch_kernel).  dy / x stay referenced until then."""
        self.wjobs.append((dy, x, partials, int(R), int(K), int(N), int(max_blocks), int(bool(with_bias))))
        self.witems.append((partials, dst, dbias, int(nblk), int(N) * int(K), int(N) if with_bias else 0, int(K), int(K), 0, 0))
        self.added += 1
        if len(self.wjobs) >= 48:
            self.flush_wgrads()

    def add_rows_wgrad(self, fields, keep, slab):
        """Queue one K-slice of a rows-stack weight gradient: `fields` = the RowsWgradJob fields (tensors as tensors, a
        (tensor, element offset) pair for a pointer into a tensor), `keep` = further tensors to hold, slab = the arguments
        of `add` for its slabs."""
        self.rjobs.append((fields, keep))
        partials, nblk, dst, n_mat, K, ldo, dbias, n_bias = slab
        self.witems.append((partials, dst, dbias, int(nblk), int(n_mat), int(n_bias), int(K), int(ldo), 0, 0))
        self.added += 1

    def defer(self, fn, keep=()):
        """Run fn() at the flush (a launch whose outputs nothing reads before the optimiser); `keep` stays referenced."""
        self.deferred.append((fn, keep))

    # The deferred launches (relation-bias backward: MFMA bound) run AFTER the batched weight gradients (HBM bound): beside
    # them the main stream walks SA2's backward and then SA1's, the most HBM-heavy run of the step — 4.49 -> 4.46 ms.
    DEFERRED_LAST = os.environ.get("VLP3D_DEFERRED_LAST", "1") != "0"

    def _run_deferred(self):
        for fn, _keep in self.deferred:
            fn()
        self.deferred = []

    def flush_wgrads(self):
        if not self.DEFERRED_LAST:
            self._run_deferred()
        if not self.wjobs and not self.rjobs:
            self._run_deferred()
            return
        n = len(self.wjobs) + len(self.rjobs)
        arr = (RowsWgradJob * n)()
        dev = None
        for d, (dy, x, partials, R, K, N, max_blocks, with_bias) in zip(arr, self.wjobs):
            dev = dy.device
            d.G, d.X, d.partials = dy.data_ptr(), x.data_ptr(), partials.data_ptr()
            d.ldg, d.lda, d.R, d.K, d.N, d.max_blocks, d.with_bias = N, K, R, K, N, max_blocks, with_bias
            d.x_bf16 = int(x.dtype == torch.bfloat16)  # an attention core's bf16 rows as the layer's input
        for d, (fields, _keep) in zip(arr[len(self.wjobs):], self.rjobs):
            for k, v in fields.items():
                if isinstance(v, tuple):  # (tensor, element offset)
                    v = v[0].data_ptr() + v[0].element_size() * v[1]
                elif torch.is_tensor(v):
                    dev = v.device
                    v = v.data_ptr()
                setattr(d, k, v)
        with torch.cuda.device(dev):
            _check(load().vlp3d_rows_wgrad_batch(ctypes.cast(arr, ctypes.c_void_p), n, _stream()), "vlp3d_rows_wgrad_batch")
        self.items.extend(self.witems)
        self.wjobs, self.rjobs, self.witems = [], [], []
        self._run_deferred()

    def flush(self, everything=True):
        """Sum the slabs that are due; everything=True (end of backward) first runs the queued linear weight gradients."""
        if everything:
            self.flush_wgrads()
        while self.items:
            chunk, self.items = self.items[:40], self.items[40:]
            arr = (SlabReduceDesc * len(chunk))()
            for d, (partials, dst, dbias, nblk, n_mat, n_bias, K, ldo, ncol_out, rot) in zip(arr, chunk):
                d.partials, d.dst, d.dbias = partials.data_ptr(), dst.data_ptr(), (dbias.data_ptr() if dbias is not None else None)
                d.nblk, d.n_mat, d.n_bias, d.K, d.ldo, d.ncol_out, d.rot = nblk, n_mat, n_bias, K, ldo, ncol_out, rot
            dev = chunk[0][0].device
            with torch.cuda.device(dev):
                _check(load().vlp3d_slab_reduce_batch(ctypes.cast(arr, ctypes.c_void_p), len(chunk), _stream()),
                       "vlp3d_slab_reduce_batch")
        self.pending = 0


_slab_queue = None  # process-wide on purpose: autograd runs backward on its own thread


def slab_queue():
    return _slab_queue


def reduce_slabs(partials, nblk, dst, n_mat, K, ldo, dbias=None, n_bias=0, ncol_out=0, rot=0):
    """Sum one launch's slabs: through the active queue (deferred) or right away."""
    q = _slab_queue
    if q is None:
        q = SlabReduceQueue()
        q.add(partials, nblk, dst, n_mat, K, ldo, dbias, n_bias, ncol_out, rot)
        q.flush()
    else:
        q.add(partials, nblk, dst, n_mat, K, ldo, dbias, n_bias, ncol_out, rot)


class deferred_slab_reduce:
    """Context: weight gradients produced inside are completed at exit (one batched slab sum)."""

    def __enter__(self):
        global _slab_queue
        self._outer = _slab_queue
        _slab_queue = SlabReduceQueue() if self._outer is None else self._outer
        return _slab_queue

    def __exit__(self, *exc):
        global _slab_queue
        q = _slab_queue
        _slab_queue = self._outer
        if self._outer is None and exc[0] is None:
            q.flush()
        return False


def call(name, *args):
    """Raw checked call of a C entry point on torch's current stream (stream appended automatically).
    Tensors are passed as pointers, None as NULL; ints/floats as they are."""
    conv = [(_opt(a) if (a is None or isinstance(a, torch.Tensor)) else a) for a in args]
    lib = _geom() if name in GEOM_ENTRY_POINTS else load()
    _check(getattr(lib, name)(*conv, _stream()), name)


# ---- in-step kernel timing (bench.py: roofline.ms) ---------------------------------------------------------------------
class Stamps:
    """While active, every call of one of the named C entry points is bracketed by two one-thread stamp kernels
    (csrc/hwprobe.hip) writing the 100 MHz device clock into consecutive slots of a device buffer — also when the calls
    are being captured into a HIP graph: replaying the graph then refreshes the slots, and `durations()` gives each
    stamped launch's duration INSIDE the step (its neighbours running, caches in their in-step state), not in isolation.
    A bracket reads  gap + duration + gap  (stamp -> kernel -> stamp); `back_to_back()` records one gap (two stamps with
    nothing between), so duration = bracket - 2 x gap.  Used on an instrumented copy of the step after bench.py's timed
    region; tools/stamp_vs_rocprof.py compares the result with rocprofv3's per-dispatch durations."""

    TICK_US = 0.01

    def __init__(self, names, device, capacity=512):
        self.names = set(names)
        self.buf = torch.zeros((capacity,), dtype=torch.int64, device=device)
        self.log = []          # slot pair k -> (entry point, positional-argument summary)
        self._saved = {}
        self.calibrate_pending = False   # set True (e.g. at capture begin): the next call of `calibrate_on` is preceded by the calibration pairs
        self.calibrate_on = "vlp3d_sa_fwd_gather"   # first stamped kernel of the main-stream graph

    def __enter__(self):
        lib = load()
        for name in self.names:
            fn = getattr(lib, name)
            self._saved[name] = fn

            def wrapped(*args, _fn=fn, _name=name):
                if self.calibrate_pending and _name == self.calibrate_on:
                    # the calibration pairs go in front of the first stamped call, on ITS stream (so that they are captured
                    # into the same graph): one bare gap, and a bracketed empty kernel
                    self.calibrate_pending = False
                    self.back_to_back(args[-1])
                    getattr(lib, "vlp3d_probe_empty")(1, 64, ctypes.c_void_p(0), args[-1])
                k = len(self.log)
                if 2 * k + 1 >= self.buf.numel():
                    return _fn(*args)
                stream = args[-1]
                self.log.append((_name, tuple(a for a in args if isinstance(a, (int, float)))))
                lib.vlp3d_stamp(ctypes.c_void_p(self.buf.data_ptr() + 16 * k), stream)
                rc = _fn(*args)
                lib.vlp3d_stamp(ctypes.c_void_p(self.buf.data_ptr() + 16 * k + 8), stream)
                return rc
            setattr(lib, name, wrapped)
        return self

    def back_to_back(self, stream=None):
        """Two stamps with nothing between them (slot pair logged as "stamp_gap"): one dispatch gap + the stamp kernel."""
        lib = load()
        stream = _stream() if stream is None else stream
        k = len(self.log)
        self.log.append(("stamp_gap", ()))
        lib.vlp3d_stamp(ctypes.c_void_p(self.buf.data_ptr() + 16 * k), stream)
        lib.vlp3d_stamp(ctypes.c_void_p(self.buf.data_ptr() + 16 * k + 8), stream)

    def __exit__(self, *exc):
        lib = load()
        for name, fn in self._saved.items():
            setattr(lib, name, fn)
        self._saved = {}
        return False

    def durations(self):
        """[(entry point, int/float args, microseconds)] for every stamped launch, from the slots' current contents."""
        t = self.buf.cpu().tolist()
        return [(n, a, (t[2 * k + 1] - t[2 * k]) * self.TICK_US) for k, (n, a) in enumerate(self.log)]

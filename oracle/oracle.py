"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement of the 3DVLP grounding hot path:

* the nine ``pointnet2._ext`` ops: plain C in ``pointnet2_oracle.c`` (loaded here
  through ctypes), restating lib/pointnet2/_ext_src/src/*.cu — "parity unpinned"
  (CUDA-only reference, no golden vectors; see that file's header);
* the Python-level glue and dense pieces (QueryAndGroup, SharedMLP, SA/FP
  modules, nn_distance, scaled-dot-product / multi-head attention, the
  cross-attention decoder layer, the OCC/OSC InfoNCE) restated in numpy — these
  ARE pinned: tests/test_oracle_golden.py checks them against fixtures produced
  by importing the reference's own Python modules (tests/golden/make_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvlp3d_oracle.so")
_lib = None

CONTRACT_NONE, CONTRACT_NVPTX, CONTRACT_LEFT = 0, 1, 2
DEFAULT_CONTRACT = CONTRACT_NVPTX


def build(force=False):
    """Compile pointnet2_oracle.c with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "pointnet2_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libvlp3d_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


def opt_n_threads(work_size):
    return int(lib().orc_opt_n_threads(int(work_size)))


# --------------------------------------------------------------------------
# the nine _ext ops (C restatement)
# --------------------------------------------------------------------------
def furthest_point_sampling(xyz, npoint, contract=DEFAULT_CONTRACT):
    """sampling.cpp:70-91 + sampling_gpu.cu:74-178. xyz (B,N,3) f32 -> (B,npoint) i32."""
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    temp = np.empty((B, N), np.float32)
    idx = np.zeros((B, npoint), np.int32)
    lib().orc_furthest_point_sampling(B, N, int(npoint), px, temp.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                      idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), int(contract))
    return idx


def gather_points(points, idx):
    """sampling.cpp:20-43. (B,C,N),(B,M) -> (B,C,M)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    M = idx.shape[1]
    out = np.zeros((B, C, M), np.float32)
    lib().orc_gather_points(B, C, N, M, pp, pi, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def gather_points_grad(grad_out, idx, n):
    """sampling.cpp:45-69. (B,C,M),(B,M) -> (B,C,n)."""
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M = grad_out.shape
    out = np.zeros((B, C, n), np.float32)
    lib().orc_gather_points_grad(B, C, int(n), M, pg, pi, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def ball_query(new_xyz, xyz, radius, nsample, contract=DEFAULT_CONTRACT):
    """ball_query.cpp:13-37 + ball_query_gpu.cu:14-49 (argument order of _ext.ball_query)."""
    new_xyz, pn = _f(new_xyz)
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), np.int32)
    lib().orc_ball_query(B, N, M, ctypes.c_float(radius), int(nsample), pn, px,
                         idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), int(contract))
    return idx


def group_points(points, idx):
    """group_points.cpp:17-40. (B,C,N),(B,M,S) -> (B,C,M,S)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    _, M, S = idx.shape
    out = np.zeros((B, C, M, S), np.float32)
    lib().orc_group_points(B, C, N, M, S, pp, pi, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def group_points_grad(grad_out, idx, n):
    """group_points.cpp:42-65. (B,C,M,S),(B,M,S) -> (B,C,n)."""
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M, S = grad_out.shape
    out = np.zeros((B, C, n), np.float32)
    lib().orc_group_points_grad(B, C, int(n), M, S, pg, pi, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def three_nn(unknown, known, contract=DEFAULT_CONTRACT):
    """interpolate.cpp:19-45. Returns (dist2, idx) — SQUARED distances, like _ext.three_nn."""
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((B, n, 3), np.float32)
    idx = np.zeros((B, n, 3), np.int32)
    lib().orc_three_nn(B, n, m, pu, pk, dist2.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                       idx.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), int(contract))
    return dist2, idx


def three_interpolate(points, idx, weight, contract=DEFAULT_CONTRACT):
    """interpolate.cpp:47-75. (B,C,m),(B,n,3),(B,n,3) -> (B,C,n)."""
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, m = points.shape
    n = idx.shape[1]
    out = np.zeros((B, C, n), np.float32)
    lib().orc_three_interpolate(B, C, m, n, pp, pi, pw, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                int(contract))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """True adjoint of three_interpolate (what interpolate_gpu.cu:121-148 intends)."""
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, n = grad_out.shape
    out = np.zeros((B, C, m), np.float32)
    lib().orc_three_interpolate_grad(B, C, n, int(m), pg, pi, pw, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)))
    return out


def three_interpolate_grad_asshipped(grad_out, idx, weight, m, contract=DEFAULT_CONTRACT):
    """What the reference executes (bug at interpolate.cpp:95): forward wrapper on grad_out."""
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, n = grad_out.shape
    out = np.zeros((B, C, m), np.float32)
    lib().orc_three_interpolate_grad_asshipped(B, C, n, int(m), pg, pi, pw,
                                               out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(contract))
    return out


# --------------------------------------------------------------------------
# numpy restatement of the Python-level pieces
# --------------------------------------------------------------------------
def huber_loss(error, delta=1.0):
    """utils/nn_distance.py:13-30."""
    abs_error = np.abs(error)
    quadratic = np.minimum(abs_error, np.float32(delta))
    linear = abs_error - quadratic
    return np.float32(0.5) * quadratic ** 2 + np.float32(delta) * linear


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    """utils/nn_distance.py:32-59. (B,N,C),(B,M,C) -> dist1 (B,N) f32, idx1 i64, dist2 (B,M), idx2 i64.
    Sum over C is performed left to right in fp32, like torch.sum over a size-3 dim."""
    pc1 = np.asarray(pc1, np.float32)
    pc2 = np.asarray(pc2, np.float32)
    diff = pc1[:, :, None, :] - pc2[:, None, :, :]
    if l1smooth:
        e = huber_loss(diff, delta)
    elif l1:
        e = np.abs(diff)
    else:
        e = diff * diff
    dist = e[..., 0]
    for c in range(1, e.shape[-1]):
        dist = dist + e[..., c]
    idx1 = np.argmin(dist, axis=2).astype(np.int64)
    idx2 = np.argmin(dist, axis=1).astype(np.int64)
    return dist.min(axis=2), idx1, dist.min(axis=1), idx2


def query_and_group(xyz, new_xyz, features, radius, nsample, use_xyz=True, normalize_xyz=False,
                    contract=DEFAULT_CONTRACT):
    """lib/pointnet2/pointnet2_utils.py:313-372 (sample_uniformly=False).
    Returns (new_features (B,3+C,M,S), grouped_xyz (B,3,M,S), idx)."""
    xyz = np.asarray(xyz, np.float32)
    new_xyz = np.asarray(new_xyz, np.float32)
    idx = ball_query(new_xyz, xyz, radius, nsample, contract)
    xyz_trans = np.ascontiguousarray(xyz.transpose(0, 2, 1))
    grouped_xyz = group_points(xyz_trans, idx)
    grouped_xyz = grouped_xyz - new_xyz.transpose(0, 2, 1)[..., None]
    if normalize_xyz:
        grouped_xyz = grouped_xyz / np.float32(radius)
    if features is not None:
        grouped_features = group_points(features, idx)
        new_features = np.concatenate([grouped_xyz, grouped_features], axis=1) if use_xyz else grouped_features
    else:
        new_features = grouped_xyz
    return new_features, grouped_xyz, idx


def shared_mlp(x, layers, training, eps=1e-5):
    """lib/pointnet2/pytorch_utils.py:11-36: stack of (1x1 conv, no bias) -> BatchNorm2d -> ReLU.
    x (B,C,M,S); layers = list of dicts {w (Co,Ci), gamma, beta, mean, var}. In training mode
    the batch statistics (biased variance) are used, like nn.BatchNorm2d.  fp64 accumulation."""
    x = np.asarray(x, np.float64)
    for L in layers:
        y = np.einsum("oc,bcms->boms", np.asarray(L["w"], np.float64), x)
        if "bias" in L and L["bias"] is not None:
            y = y + np.asarray(L["bias"], np.float64)[None, :, None, None]
        if L.get("gamma") is not None:
            if training:
                mean = y.mean(axis=(0, 2, 3))
                var = y.var(axis=(0, 2, 3))
            else:
                mean, var = np.asarray(L["mean"], np.float64), np.asarray(L["var"], np.float64)
            y = (y - mean[None, :, None, None]) / np.sqrt(var[None, :, None, None] + eps)
            y = y * np.asarray(L["gamma"], np.float64)[None, :, None, None] + \
                np.asarray(L["beta"], np.float64)[None, :, None, None]
        x = np.maximum(y, 0.0)
    return x.astype(np.float32)


def sa_module_votes(xyz, features, layers, npoint, radius, nsample, training, normalize_xyz=True,
                    inds=None, contract=DEFAULT_CONTRACT):
    """lib/pointnet2/pointnet2_modules.py:210-272 (pooling='max', use_xyz=True).
    Returns (new_xyz (B,npoint,3), new_features (B,Cout,npoint), inds (B,npoint) i32)."""
    xyz = np.asarray(xyz, np.float32)
    if inds is None:
        inds = furthest_point_sampling(xyz, npoint, contract)
    xyz_flipped = np.ascontiguousarray(xyz.transpose(0, 2, 1))
    new_xyz = np.ascontiguousarray(gather_points(xyz_flipped, inds).transpose(0, 2, 1))
    grouped, _, _ = query_and_group(xyz, new_xyz, features, radius, nsample, True, normalize_xyz, contract)
    y = shared_mlp(grouped, layers, training)
    return new_xyz, y.max(axis=3), inds


def fp_module(unknown, known, unknow_feats, known_feats, layers, training, contract=DEFAULT_CONTRACT):
    """lib/pointnet2/pointnet2_modules.py:371-416."""
    dist2, idx = three_nn(unknown, known, contract)
    dist = np.sqrt(dist2)  # pointnet2_utils.py:138-140
    dist_recip = np.float32(1.0) / (dist + np.float32(1e-8))
    norm = dist_recip.sum(axis=2, keepdims=True)
    weight = dist_recip / norm
    interpolated = three_interpolate(known_feats, idx, weight, contract)
    new_features = np.concatenate([interpolated, unknow_feats], axis=1) if unknow_feats is not None else interpolated
    return shared_mlp(new_features[..., None], layers, training)[..., 0]


def _linear(x, w, b=None):
    y = np.asarray(x, np.float64) @ np.asarray(w, np.float64).T
    return y if b is None else y + np.asarray(b, np.float64)


def _layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(axis=-1, keepdims=True)
    var = x.var(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * np.asarray(g, np.float64) + np.asarray(b, np.float64)


def softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def sdpa_core(q, k, v, attention_mask=None, attention_weights=None, way="add"):
    """models/transformer/attention.py:63-75 on already-projected heads.
    q (B,h,nq,dk), k (B,h,nk,dk), v (B,h,nk,dv) -> out (B,h,nq,dv), att (B,h,nq,nk). fp64."""
    q, k, v = (np.asarray(t, np.float64) for t in (q, k, v))
    att = q @ k.transpose(0, 1, 3, 2) / np.sqrt(q.shape[-1])
    if attention_weights is not None:
        if way == "mul":
            att = att * attention_weights
        elif way == "add":
            att = att + attention_weights
        else:
            raise NotImplementedError(way)
    if attention_mask is not None:
        att = np.where(np.asarray(attention_mask) == 0, -10000.0, att)
    att = softmax(att, -1)
    return att @ v, att


def scaled_dot_product_attention(P, queries, keys, values, h, attention_mask=None, attention_weights=None,
                                 way="add"):
    """models/transformer/attention.py:41-78. P: dict with fc_{q,k,v,o}.{weight,bias}."""
    b_s, nq = queries.shape[:2]
    nk = keys.shape[1]
    q = _linear(queries, P["fc_q.weight"], P["fc_q.bias"]).reshape(b_s, nq, h, -1).transpose(0, 2, 1, 3)
    k = _linear(keys, P["fc_k.weight"], P["fc_k.bias"]).reshape(b_s, nk, h, -1).transpose(0, 2, 1, 3)
    v = _linear(values, P["fc_v.weight"], P["fc_v.bias"]).reshape(b_s, nk, h, -1).transpose(0, 2, 1, 3)
    out, att = sdpa_core(q, k, v, attention_mask, attention_weights, way)
    out = out.transpose(0, 2, 1, 3).reshape(b_s, nq, -1)
    return _linear(out, P["fc_o.weight"], P["fc_o.bias"]), att


def _sub(P, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in P.items() if k.startswith(prefix)}


def multi_head_attention(P, queries, keys, values, h, attention_mask=None, attention_weights=None, way="add"):
    """models/transformer/attention.py:108-131, eval mode (dropout = identity), post-LN residual."""
    out, att = scaled_dot_product_attention(_sub(P, "attention."), queries, keys, values, h, attention_mask,
                                            attention_weights, way)
    return _layer_norm(np.asarray(queries, np.float64) + out, P["layer_norm.weight"], P["layer_norm.bias"]), att


def cross_attention_decoder_layer(P, query, key, value, h=4, src_mask=None, src_trg_mask=None):
    """models/transformer/mmattention.py:68-86, eval mode."""
    x, _ = multi_head_attention(_sub(P, "self_attention."), query, query, query, h, attention_mask=src_mask)
    x, _ = multi_head_attention(_sub(P, "enc_dec_attention."), x, key, value, h, attention_mask=src_trg_mask)
    _x = x
    y = np.maximum(_linear(x, P["ffn.linear1.weight"], P["ffn.linear1.bias"]), 0.0)
    y = _linear(y, P["ffn.linear2.weight"], P["ffn.linear2.bias"])
    return _layer_norm(y + _x, P["norm.weight"], P["norm.bias"])


def soft_cross_entropy(inputs, target):
    """models/constrast_module/constrast_module.py:18-21 — MEAN over all elements."""
    inputs = np.asarray(inputs, np.float64)
    x = inputs - inputs.max(axis=-1, keepdims=True)
    logsm = x - np.log(np.exp(x).sum(axis=-1, keepdims=True))
    return float(np.mean(-logsm * np.asarray(target, np.float64)))


def nce_loss(logits, iou_matrix):
    """constrast_module.py:34-37 (tau unused: :32-33 commented out)."""
    logits = np.asarray(logits, np.float64)
    iou_matrix = np.asarray(iou_matrix, np.float64)
    loss_v = soft_cross_entropy(logits, iou_matrix)
    # SoftCrossEntropy(logits.t(), iou_matrix): elementwise product broadcasts (P,1)x(1,P) for OCC.
    lt = logits.T
    x = lt - lt.max(axis=-1, keepdims=True)
    logsm = x - np.log(np.exp(x).sum(axis=-1, keepdims=True))
    loss_t = float(np.mean(-logsm * iou_matrix))
    return (loss_v + loss_t) / 2


def box3d_iou_axis_aligned(center1, size1, center2, size2):
    """Closed-form IoU of axis-aligned boxes; value-identical to the `iou` of
    utils/box_util.py:488-529 (box3d_diou_batch_tensor). Replaces pytorch3d box3d_overlap
    (constrast_module.py:105) whose inputs are always axis aligned (create_box_batch :9-15)."""
    c1, s1, c2, s2 = (np.asarray(t, np.float64) for t in (center1, size1, center2, size2))
    lo = np.maximum(c1 - s1 / 2, c2 - s2 / 2)
    hi = np.minimum(c1 + s1 / 2, c2 + s2 / 2)
    inter = np.prod(np.clip(hi - lo, 0, None), axis=-1)
    vol1, vol2 = np.prod(s1, axis=-1), np.prod(s2, axis=-1)
    return inter / (vol1 + vol2 - inter)


def _normalize(x, eps=1e-12):
    n = np.sqrt((x * x).sum(axis=-1, keepdims=True))
    return x / np.maximum(n, eps)


def contrast_losses(W, pred_center, pred_size, bbox_feature, objectness_scores, gt_center, gt_size, lang_emb,
                    lang_num):
    """constrast_module.py:53-131 for epoch >= 50.  W: pc_proj / text_proj / pc_proj_iou weights (128,128).
    gt_center/gt_size (B,L,3) are the decoded GT boxes (param2obb_batch_tensor output).
    Returns (lang_con_loss, iou_con_loss)."""
    B = pred_center.shape[0]
    lang_emb = np.asarray(lang_emb, np.float64).reshape(B, -1, lang_emb.shape[-1])
    occ = 0.0
    osc = 0.0
    for i in range(B):
        obj = np.where(np.argmax(objectness_scores[i], axis=1) == 1)[0]
        feats = np.asarray(bbox_feature[i], np.float64)[obj]
        for j in range(int(lang_num[i])):
            ious = box3d_iou_axis_aligned(gt_center[i, j][None], gt_size[i, j][None] + 1e-2,
                                          pred_center[i][obj], pred_size[i][obj])
            mask = (ious > 0.25).astype(np.float64)
            t = _normalize(lang_emb[i, j][None] @ np.asarray(W["text_proj"], np.float64).T)
            p = _normalize(feats @ np.asarray(W["pc_proj"], np.float64).T)
            occ += nce_loss(t @ p.T, mask[None, :])
            pi = _normalize(feats @ np.asarray(W["pc_proj_iou"], np.float64).T)
            osc += nce_loss(pi @ pi.T, np.outer(mask, mask))
    return occ / B, osc / B

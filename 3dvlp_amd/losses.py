"""The training loss of the grounding path — SURVEY.md §8f-1 — with the reference's function names.

  compute_vote_loss / compute_objectness_loss      lib/loss_helper/loss_detection.py:24-113
  recover_assigned_gt_bboxes / compute_box_loss /
  compute_box_and_sem_cls_loss                     lib/loss_helper/loss_detection.py:116-258
  box3d_diou_batch_tensor                          utils/box_util.py:488-529
  SoftmaxRankingLoss                               lib/loss_helper/loss.py:6-17
  compute_diou_loss                                lib/loss_helper/loss_grounding.py:129-365
  get_joint_loss                                   lib/loss_helper/loss_joint.py:26-227

Two implementations of the same arithmetic:
  * ``impl="hip"`` (default, the product path): csrc/joint_loss.hip — one forward kernel + a finalize kernel and one
    backward kernel for everything except the OCC/OSC terms (which the contrast module already computed on device).
    CUDA tensors only; raises on CPU tensors like every other op of the package.
  * ``impl="torch"``: the same formulas as batched torch ops (no Python loop over scenes / sentences and no host sync,
    unlike the reference, but term by term the same math).  An explicit opt-in used by the tests as the bridge between
    the oracle's literal loops (CPU) and the kernels (GPU); never selected automatically.

Deliberate, documented differences from the reference:
  * the per-(scene, sentence) Python loops with `.cpu()` syncs are batched;
  * out-of-scope switches (`use_reg_head`, `use_kl_loss`, `use_attr_loss`, `use_vote_weight`,
    `use_mlm`, `orientation`, `distance`) raise NotImplementedError when turned on;
  * the language-classification term reads `lang_scores` of the (out-of-scope) language encoder: it is included when the
    data_dict carries `lang_scores` + `object_cat_list`, else reported as zero.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

from . import _lib as _ext
from .nn_distance import huber_loss

FAR_THRESHOLD = 0.3   # loss_detection.py:19-22 (both 0.3: every proposal is either near or far)
NEAR_THRESHOLD = 0.3
GT_VOTE_FACTOR = 3
OBJECTNESS_CLS_WEIGHTS = [0.2, 0.8]

DEFAULT_IMPL = "hip"
_CONST = {}


def _const(key, make, device):
    k = (key, str(device))
    if k not in _CONST:
        _CONST[k] = make().to(device)
    return _CONST[k]


def _mean_size(config, device):
    """config.mean_size_arr as a device tensor, cached ON the config object (per device).  Round 2 keyed a module-level
    cache by id(config): a freed config's id is handed to the next one, which then decoded sizes with the previous
    config's table (found by the direct HIP-vs-oracle tests of round 3)."""
    cache = config.__dict__.setdefault("_vlp3d_mean_size", {}) if hasattr(config, "__dict__") else {}
    k = str(device)
    src = config.mean_size_arr
    hit = cache.get(k)
    if hit is None or hit[0] is not src:
        hit = (src, torch.as_tensor(np.asarray(src, np.float32)).to(device))
        cache[k] = hit
    return hit[1]


class SoftmaxRankingLoss(nn.Module):
    """loss.py:6-17."""

    def forward(self, inputs, targets):
        assert inputs.shape == targets.shape
        probs = F.softmax(inputs + 1e-8, dim=-1)
        return -torch.sum(torch.log(probs + 1e-8) * targets, dim=-1).mean()


def box3d_diou_batch_tensor(center1, size1, center2, size2):
    """utils/box_util.py:488-529, broadcasting over leading dims -> (iou, diou)."""
    lo1, hi1 = center1 - size1 / 2, center1 + size1 / 2
    lo2, hi2 = center2 - size2 / 2, center2 + size2 / 2
    area1 = size1[..., 0] * size1[..., 1] * size1[..., 2]
    area2 = size2[..., 0] * size2[..., 1] * size2[..., 2]
    e = torch.clamp(torch.min(hi1, hi2) - torch.max(lo1, lo2), min=0)
    inter = e[..., 0] * e[..., 1] * e[..., 2]
    iou = inter / (area1 + area2 - inter)
    dc = center1 - center2
    inter_diag = dc[..., 0] ** 2 + dc[..., 1] ** 2 + dc[..., 2] ** 2
    o = torch.clamp(torch.max(hi1, hi2) - torch.min(lo1, lo2), min=0)
    outer_diag = o[..., 0] ** 2 + o[..., 1] ** 2 + o[..., 2] ** 2
    diou = torch.clamp(iou - 1.5 * inter_diag / outer_diag, min=-1, max=1)
    return iou, diou


# ---------------------------------------------------------------------------------------------------------------------
# impl="torch": batched op-by-op form
# ---------------------------------------------------------------------------------------------------------------------
def compute_vote_loss(data_dict):
    B, S = data_dict["seed_xyz"].shape[:2]
    seed_inds = data_dict["seed_inds"].long()
    mask = torch.gather(data_dict["vote_label_mask"], 1, seed_inds).float()
    gt = torch.gather(data_dict["vote_label"], 1, seed_inds.unsqueeze(-1).expand(-1, -1, 3 * GT_VOTE_FACTOR))
    gt = (gt + data_dict["seed_xyz"].repeat(1, 1, 3)).reshape(B * S, GT_VOTE_FACTOR, 3)
    vote = data_dict["vote_xyz"].reshape(B * S, -1, 3)
    dist = (vote[:, :, None, :] - gt[:, None, :, :]).abs().sum(-1)       # nn_distance(..., l1=True), dense
    votes_dist = dist.min(dim=1)[0].min(dim=1)[0].view(B, S)
    return torch.sum(votes_dist * mask) / (torch.sum(mask) + 1e-6)


def compute_objectness_loss(data_dict):
    agg = data_dict["aggregated_vote_xyz"]
    gt_center = data_dict["center_label"][:, :, 0:3]
    diff = agg.detach()[:, :, None, :] - gt_center[:, None, :, :]
    sq = diff * diff
    dist1, ind1 = ((sq[..., 0] + sq[..., 1]) + sq[..., 2]).min(dim=2)   # nn_distance: squared L2, first minimum
    euc = torch.sqrt(dist1 + 1e-6)
    label = (euc < NEAR_THRESHOLD).long()
    mask = ((euc < NEAR_THRESHOLD) | (euc > FAR_THRESHOLD)).float()
    w = _const("objw", lambda: torch.tensor(OBJECTNESS_CLS_WEIGHTS), agg.device)
    ce = F.cross_entropy(data_dict["objectness_scores"].float().transpose(2, 1), label, weight=w, reduction="none")
    return torch.sum(ce * mask) / (torch.sum(mask) + 1e-6), label, mask, ind1


def recover_assigned_gt_bboxes(data_dict, config, object_assignment):
    nh = config.num_heading_bin
    agg = data_dict["aggregated_vote_xyz"]
    dev = agg.device
    B, K = object_assignment.shape
    a3 = object_assignment.unsqueeze(-1).expand(-1, -1, 3)
    gt_center = torch.gather(data_dict["center_label"][:, :, 0:3], 1, a3)
    hcl = torch.gather(data_dict["heading_class_label"], 1, object_assignment)
    hrl = torch.gather(data_dict["heading_residual_label"], 1, object_assignment)
    if nh != 1:
        gt_heading = hcl.float() * ((2 * np.pi) / float(nh)) + hrl
    else:
        gt_heading = torch.zeros((B, K), device=dev)
    scl = torch.gather(data_dict["size_class_label"], 1, object_assignment)
    srl = torch.gather(data_dict["size_residual_label"], 1, a3)
    mean = _mean_size(config, dev)
    gt_size = mean[scl] + srl
    half = gt_size / 2
    off = agg - gt_center                                   # carries gradient to the vote centres, as in the reference
    c, s = torch.cos(-gt_heading), torch.sin(-gt_heading)   # row vector @ rotz_batch_pytorch(-heading)
    rot = torch.stack([off[..., 0] * c + off[..., 1] * s, -off[..., 0] * s + off[..., 1] * c, off[..., 2]], -1)
    gt_distance = torch.cat([half + rot, half - rot], dim=2)
    return gt_center, hcl, hrl, gt_heading, gt_distance, gt_size


def compute_box_and_sem_cls_loss(data_dict, config):
    nh = config.num_heading_bin
    assign = data_dict["object_assignment"]
    gt_center, hcl, hrl, gt_heading, gt_distance, _ = recover_assigned_gt_bboxes(data_dict, config, assign)
    data_dict["gt_assigned_center"], data_dict["gt_assigned_heading_class"] = gt_center, hcl
    data_dict["gt_assigned_heading_residual"], data_dict["gt_assigned_heading"] = hrl, gt_heading
    data_dict["gt_assigned_distance"] = gt_distance
    lab = data_dict["objectness_label"].float()
    den = torch.sum(lab) + 1e-6
    hc = F.cross_entropy(data_dict["heading_scores"].transpose(2, 1), hcl, reduction="none")
    hc = torch.sum(hc * lab) / den
    onehot = F.one_hot(hcl, nh).float()
    res = torch.sum(data_dict["heading_residuals_normalized"] * onehot, -1) - hrl / (np.pi / nh)
    hr = torch.sum(huber_loss(res, delta=1.0) * lab) / den
    dist = torch.mean(huber_loss(data_dict["rois"] - gt_distance, delta=0.15), -1)
    dl = torch.sum(dist * lab) / den
    sem_label = torch.gather(data_dict["sem_cls_label"], 1, assign)
    sem = F.cross_entropy(data_dict["sem_cls_scores"].transpose(2, 1), sem_label, reduction="none")
    sem = torch.sum(sem * lab) / den
    return hc, hr, dl, sem


def _train_gate(data_dict, device, no_reference=False):
    """`istrain == 1 and not no_reference and random < 0.5` (loss_grounding.py:249) as a device boolean."""
    if data_dict["istrain"][0] != 1 or no_reference:
        return torch.zeros((), dtype=torch.bool, device=device)
    return torch.as_tensor(data_dict["random"], device=device) < 0.5


def compute_diou_loss(data_dict, config, no_reference=False, use_reg_head=False, use_kl_loss=False, debug=False):
    """-> (data_dict, ref_loss, cluster_preds (B,L,K), cluster_labels (B,L,K)); stores diou_loss and the IoU rates."""
    if use_reg_head or use_kl_loss:
        raise NotImplementedError("use_reg_head / use_kl_loss are outside the grounding hot path (SURVEY.md §8)")
    pred_center, pred_size = data_dict["pred_center"], data_dict["pred_size"]
    dev = pred_center.device
    gt_center = data_dict["ref_center_label_list"].detach()[..., 0:3]
    B, K = pred_center.shape[:2]
    L = gt_center.shape[1]
    lang_num = data_dict["lang_num"]
    mean = _mean_size(config, dev)
    gt_size = mean[data_dict["ref_size_class_label_list"]] + data_dict["ref_size_residual_label_list"]
    cluster_preds = data_dict["cluster_ref"].reshape(B, L, K)
    iou, diou = box3d_diou_batch_tensor(pred_center[:, None, :, :], pred_size[:, None, :, :],
                                        gt_center[:, :, None, :], gt_size[:, :, None, :].float())
    iou_np = iou.detach()
    obj = data_dict["objectness_scores"].max(2)[1].float()
    iou_m = torch.where(_train_gate(data_dict, dev, no_reference), iou_np * obj[:, None, :], iou_np)
    row_ok = torch.arange(L, device=dev)[None, :] < lang_num[:, None]            # j < lang_num[i]
    max_iou, ind = iou_np.max(-1)
    valid = row_ok & (max_iou >= 0.25)
    amax = iou_m.argmax(-1)
    labels = F.one_hot(ind, K).float() * valid[..., None]
    if data_dict["epoch"] < 50:
        sm = iou_m >= 0.25
        cnt = sm.sum(-1)
        multi = cnt >= 2
        smooth = torch.where(multi[..., None] & sm, (0.05 / (cnt - 1).clamp(min=1).float())[..., None], 0.0)
        top = torch.where(multi, 0.95, 1.0)
        smooth = smooth.scatter(2, amax[..., None], top[..., None])
    else:
        smooth = F.one_hot(amax, K).float()
    smooth = (smooth * valid[..., None]).detach()
    probs = F.softmax(cluster_preds + 1e-8, dim=-1)
    rows = -torch.sum(torch.log(probs + 1e-8) * smooth, dim=-1) * row_ok             # (B,L)
    loss = (rows.sum(1) / lang_num.float()).sum() / B                                  # mean over the scene's rows
    diou_loss = torch.sum((1 - diou) * smooth * row_ok[..., None]) / B
    tot = torch.sum(lang_num).float()
    data_dict["max_iou_rate_0.25"] = valid.sum().float() / tot
    data_dict["max_iou_rate_0.5"] = (row_ok & (max_iou >= 0.5)).sum().float() / tot
    data_dict["diou_loss"] = diou_loss
    return data_dict, loss, cluster_preds, labels.detach()


def compute_lang_classification_loss(data_dict):
    """loss_grounding.py:476-487, batched."""
    cats = data_dict["object_cat_list"]
    B, L = cats.shape[:2]
    scores = data_dict["lang_scores"].reshape(B, L, -1)
    ce = F.cross_entropy(scores.transpose(2, 1), cats, reduction="none")
    ok = (torch.arange(L, device=ce.device)[None, :] < data_dict["lang_num"][:, None]).float()
    return ((ce * ok).sum(1) / data_dict["lang_num"].float()).sum() / B


# ---------------------------------------------------------------------------------------------------------------------
# impl="hip": fused kernels
# ---------------------------------------------------------------------------------------------------------------------
OUT_NAMES = ("vote_loss", "objectness_loss", "heading_cls_loss", "heading_reg_loss", "size_distance_loss",
             "sem_cls_loss", "box_loss", "ref_loss", "diou_loss", "loss", "pos_ratio", "neg_ratio", "obj_acc",
             "max_iou_rate_0.25", "max_iou_rate_0.5")


class _JointLossCore(Function):
    """csrc/joint_loss.hip.  out = the OUT_NAMES scalars; only out[9] (the weighted total) carries gradient, to
    vote_xyz, objectness_scores, heading_scores, heading_residuals_normalized, rois, sem_cls_scores,
    aggregated_vote_xyz, pred_center, pred_size and cluster_ref."""

    @staticmethod
    def forward(ctx, vote_xyz, obj_scores, heading_scores, heading_res_norm, rois, sem_scores, agg_xyz, pred_center,
                pred_size, cluster_ref, labels, cfg):
        cf = lambda t: t.contiguous().float()
        diff = [cf(t) for t in (vote_xyz, obj_scores, heading_scores, heading_res_norm, rois, sem_scores, agg_xyz,
                                pred_center, pred_size, cluster_ref)]
        (seed_xyz, seed_inds, vote_label, vote_mask, center_label, hcl, hrl, scl, srl, sem_label, ref_center, ref_size,
         lang_num, coin, mean_size) = labels
        B, S = seed_inds.shape
        N, K, G, L = vote_mask.shape[1], agg_xyz.shape[1], center_label.shape[1], ref_center.shape[1]
        NH, NC = heading_scores.shape[2], sem_scores.shape[2]
        dims = (B, S, N, K, G, L, NH, NC)
        dev = vote_xyz.device
        nrow = int(_ext.load().vlp3d_joint_loss_rows(B, S, K, L))
        part = torch.empty((nrow, 16), dtype=torch.float64, device=dev)
        sums = torch.empty((16,), dtype=torch.float64, device=dev)
        out = torch.empty((len(OUT_NAMES),), dtype=torch.float32, device=dev)
        assign = torch.empty((B, K), dtype=torch.int32, device=dev)       # object_assignment
        objlab = torch.empty((B, K), dtype=torch.int32, device=dev)       # objectness_label
        rowinfo = torch.empty((B, L, 4), dtype=torch.int32, device=dev)   # valid, argmax(iou), argmax(masked iou), cnt
        fixed = (seed_xyz, seed_inds, vote_label, vote_mask, center_label, hcl, hrl, scl, srl, sem_label, ref_center,
                 ref_size, lang_num, coin, mean_size)
        _ext.call("vlp3d_joint_loss_fwd", *diff, *fixed, *dims, *cfg, part, sums, out, assign, objlab, rowinfo)
        ctx.save_for_backward(*diff, *fixed, sums, assign, objlab, rowinfo)
        ctx.dims, ctx.cfg = dims, cfg
        ctx.mark_non_differentiable(assign, objlab, rowinfo)
        ctx.set_materialize_grads(False)   # no zero-filled cotangents for the three index outputs (three fill launches)
        return out, assign, objlab, rowinfo

    @staticmethod
    def backward(ctx, gout, _a, _b, _c):
        sv = ctx.saved_tensors
        diff, fixed, (sums, assign, objlab, rowinfo) = sv[:10], sv[10:25], sv[25:]
        if gout is None:
            return (None,) * 12
        g = gout[9:10].contiguous()  # only the total is differentiable (the components are reporting values)
        grads = [torch.empty_like(t) for t in diff]
        _ext.call("vlp3d_joint_loss_bwd", *diff, *fixed, *ctx.dims, *ctx.cfg, sums, assign, objlab, rowinfo, g, *grads)
        return (*grads, None, None)


class _LossTail(Function):
    """loss_joint.py:204-223 in one launch each way (csrc/glue.hip loss_tail): total = ((core[9] + 0.3 lang) + (0.5 lang_con +
    2.5 iou_con)) + answer + caption, in the reference's fp32 order; returns (total, con_loss).  Replaces select / mul / mul /
    add / add forward and ones-like / mul / mul / zeros / copy backward (SelectBackward of the core's total included).  Absent
    terms are None.  con_loss is a reporting value: a cotangent on it is not supported (raises)."""

    W_LANG, W_LCON, W_ICON = 0.3, 0.5, 2.5

    @staticmethod
    def forward(ctx, core, lang, lcon, icon, ans, cap):
        f = lambda t: None if t is None else t.reshape(1).float()   # views: 0-dim / (1,) fp32 scalars on the device
        total = torch.empty((2,), dtype=torch.float32, device=core.device)
        _ext.call("vlp3d_loss_tail_fwd", core[9:10], f(lang), f(lcon), f(icon), f(ans), f(cap), _LossTail.W_LANG,
                  _LossTail.W_LCON, _LossTail.W_ICON, total)
        ctx.shapes = [None if t is None else (t.shape, t.dtype) for t in (lang, lcon, icon, ans, cap)]
        ctx.ncore = core.shape[0]
        ctx.set_materialize_grads(False)
        return total[0], total[1]

    @staticmethod
    def backward(ctx, g, g_con):
        if g_con is not None:
            raise RuntimeError("_LossTail: con_loss is a reporting value (take 0.5 * lang_con_loss + 2.5 * iou_con_loss instead)")
        if g is None:
            return (None,) * 6
        n = ctx.ncore
        d = torch.empty((n + 4,), dtype=torch.float32, device=g.device)
        _ext.call("vlp3d_loss_tail_bwd", g.reshape(1).contiguous().float(), n, 9, _LossTail.W_LANG, _LossTail.W_LCON,
                  _LossTail.W_ICON, d)
        outs = [d[:n]]   # d(core) = g e_9
        for sh, k in zip(ctx.shapes, (1, 2, 3, 0, 0)):
            outs.append(None if sh is None else d[n + k].reshape(sh[0]).to(sh[1]))
        return tuple(outs)


def _labels(data_dict, config, device):
    d = data_dict
    f = lambda t: t.contiguous().float()
    i32 = lambda t: t.contiguous().to(torch.int32)
    mean = _mean_size(config, device)
    ref_size = d["k/ref_size"] if "k/ref_size" in d else \
        (mean[d["ref_size_class_label_list"]] + d["ref_size_residual_label_list"]).float().contiguous()
    k = lambda name, conv: d["k/" + name] if ("k/" + name) in d else conv(d[name])  # loader-prepared form when present
    if d["istrain"][0] == 1 and "random" in d:
        coin = torch.as_tensor(d["random"], device=device, dtype=torch.float32).reshape(1)
    else:
        coin = _const("coin_off", lambda: torch.ones(1), device)  # >= 0.5: no objectness gating of the IoUs
    return (f(d["seed_xyz"]), i32(d["seed_inds"]), f(d["vote_label"]), k("vote_label_mask", f),
            f(d["center_label"][:, :, 0:3]), k("heading_class_label", i32), f(d["heading_residual_label"]),
            k("size_class_label", i32), f(d["size_residual_label"]), k("sem_cls_label", i32),
            f(d["ref_center_label_list"][..., 0:3]), ref_size, k("lang_num", i32), coin, mean.contiguous())


class _BceLogits(torch.autograd.Function):
    """sum(binary_cross_entropy_with_logits(x, t)) / rows: one reduction launch pair forward, one launch backward
    (csrc/glue.hip) instead of the op's seven element-wise / reduction launches each way."""

    @staticmethod
    def forward(ctx, x, t):
        x, t = x.contiguous(), t.contiguous().float()
        rows, cols = x.shape[0], x.numel() // x.shape[0]
        part = torch.empty((int(_ext.load().vlp3d_bce_logits_blocks(x.numel())),), dtype=torch.float64, device=x.device)
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
        _ext.call("vlp3d_bce_logits_fwd", x, t, rows, cols, part, out)
        ctx.save_for_backward(x, t)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        dx = torch.empty_like(x)
        _ext.call("vlp3d_bce_logits_bwd", x, t, x.shape[0], x.numel() // x.shape[0], g.reshape(1).contiguous().float(), dx)
        return dx, None


def compute_answer_classification_loss(data_dict):
    """lib/loss_helper/loss_answering.py:2-16: soft-score targets -> summed BCE-with-logits per question; class index
    targets -> cross entropy."""
    scores = data_dict["answer_scores"]
    if "answer_cat_scores" in data_dict:
        if scores.is_cuda and scores.dtype == torch.float32:
            return _BceLogits.apply(scores, data_dict["answer_cat_scores"])
        return F.binary_cross_entropy_with_logits(scores, data_dict["answer_cat_scores"].to(scores.dtype),
                                                  reduction="sum") / scores.shape[0]
    return F.cross_entropy(scores, data_dict["answer_cat"])


class _Args:
    """Defaults of scripts/joint_scripts/train_3dvlp.py:590-770 for the switches get_joint_loss reads, with run.sh:1's
    flags (--use_con --use_diou_loss) on."""
    use_reg_head = use_kl_loss = debug = use_attr_loss = use_vote_weight = use_answer = use_mlm = False
    use_diou_loss = use_con = True


def get_joint_loss(args, data_dict, device=None, config=None, weights=None, pad_token_id=None, detection=True,
                   caption=False, reference=True, use_lang_classifier=True, orientation=False, distance=False,
                   num_bins=None, num_ground_epoch=50, tokenizer=None, impl=None):
    """loss_joint.py:26-227 (same signature; `impl` is the only addition).  Writes the reference's data_dict keys and
    returns data_dict with data_dict["loss"]."""
    args = _Args if args is None else args
    for flag in ("use_reg_head", "use_kl_loss", "use_attr_loss", "use_vote_weight", "use_mlm"):
        if getattr(args, flag, False):
            raise NotImplementedError(flag + " is outside the grounding hot path (SURVEY.md §8)")
    if orientation or distance or not detection or not reference:
        raise NotImplementedError("only detection + reference (the run.sh:1 configuration; + caption for BASELINE cfg4) is on "
                                  "the grounding hot path")
    impl = impl or DEFAULT_IMPL
    d = data_dict
    dev = d["vote_xyz"].device
    w_ref = 0.3 if d["epoch"] < 50 else 1.0
    w_diou = 0.3 if getattr(args, "use_diou_loss", False) else 0.0
    if impl == "hip":
        if not d["vote_xyz"].is_cuda:
            raise RuntimeError("CPU not supported (impl='hip'); pass impl='torch' explicitly for host-side tests")
        cfg = (float(NEAR_THRESHOLD), float(FAR_THRESHOLD), float(OBJECTNESS_CLS_WEIGHTS[0]),
               float(OBJECTNESS_CLS_WEIGHTS[1]), float(w_ref), float(w_diou), int(d["epoch"] < 50))
        out, assign, objlab, rowinfo = _JointLossCore.apply(
            d["vote_xyz"], d["objectness_scores"], d["heading_scores"], d["heading_residuals_normalized"], d["rois"],
            d["sem_cls_scores"], d["aggregated_vote_xyz"], d["pred_center"], d["pred_size"], d["cluster_ref"],
            _labels(d, config, dev), cfg)
        comp = out.detach()
        for i, name in enumerate(OUT_NAMES):
            if name != "loss":
                d[name] = comp[i]
        B, L = rowinfo.shape[:2]
        K = d["cluster_ref"].shape[-1]
        # reporting tensors in the reference's dtypes; cluster_labels = hard one-hot of the best-IoU proposal, zero rows
        # where no proposal reaches 0.25 (one launch: csrc/joint_loss.hip jl_report)
        a64 = torch.empty((B, K), dtype=torch.int64, device=dev)
        l64 = torch.empty((B, K), dtype=torch.int64, device=dev)
        msk = torch.empty((B, K), dtype=torch.float32, device=dev)
        cl = torch.empty((B, L, K), dtype=torch.float32, device=dev)
        _ext.call("vlp3d_joint_loss_report", assign, objlab, rowinfo, B, K, L, a64, l64, msk, cl)
        d["object_assignment"], d["objectness_label"], d["objectness_mask"], d["cluster_labels"] = a64, l64, msk, cl
        loss = None   # the total is formed by _LossTail below, from the core's vector and the optional terms
        core_out = out
    elif impl == "torch":
        vote_loss = compute_vote_loss(d)
        obj_loss, label, mask, assign = compute_objectness_loss(d)
        total = float(label.numel())
        d["objectness_label"], d["objectness_mask"], d["object_assignment"] = label, mask, assign
        d["pos_ratio"] = torch.sum(label.float()) / total
        d["neg_ratio"] = torch.sum(mask) / total - d["pos_ratio"]
        hc, hr, dl, sem = compute_box_and_sem_cls_loss(d, config)
        box_loss = 0.1 * hc + hr + 0.1 * sem + 20 * dl
        pred = torch.argmax(d["objectness_scores"], 2)
        d["obj_acc"] = torch.sum((pred == label).float() * mask) / (torch.sum(mask) + 1e-6)
        d["vote_loss"], d["objectness_loss"], d["heading_cls_loss"], d["heading_reg_loss"] = vote_loss, obj_loss, hc, hr
        d["size_distance_loss"], d["sem_cls_loss"], d["box_loss"] = dl, sem, box_loss
        d, ref_loss, _, cluster_labels = compute_diou_loss(d, config, use_reg_head=False, use_kl_loss=False)
        d["cluster_labels"], d["ref_loss"] = cluster_labels, ref_loss
        loss = 10 * (vote_loss + 0.1 * obj_loss + box_loss) + w_ref * ref_loss + w_diou * d["diou_loss"]
    else:
        raise ValueError(impl)
    fused_tail = impl == "hip"
    terms = dict(lang=None, lcon=None, icon=None, ans=None, cap=None)
    if use_lang_classifier and "lang_scores" in d and "object_cat_list" in d:
        d["lang_loss"] = compute_lang_classification_loss(d)
        terms["lang"] = d["lang_loss"]
    else:
        d["lang_loss"] = _const("zero", lambda: torch.zeros(()), dev)
    if getattr(args, "use_con", False):
        if d["epoch"] >= 50:
            terms["lcon"], terms["icon"] = d["lang_con_loss"], d["iou_con_loss"]
            if not fused_tail:
                d["con_loss"] = 0.5 * d["lang_con_loss"] + 2.5 * d["iou_con_loss"]   # loss_joint.py:208
    else:
        d["con_loss"] = torch.zeros(1)
    if getattr(args, "use_answer", False):  # loss_joint.py:118-119, 219-220 (the ScanQA + grounding joint task, cfg5)
        d["answer_loss"] = compute_answer_classification_loss(d)
        terms["ans"] = d["answer_loss"]
    zero_keys = ["ori_loss", "ori_acc", "dist_loss", "mlm_loss"]
    if caption:  # loss_joint.py:122-127, 222-223 (Scan2Cap head on the shared proposal features, BASELINE cfg4)
        from .caption import compute_cap_loss
        d["cap_loss"], d["cap_acc"] = compute_cap_loss(d, pad_token_id=0 if pad_token_id is None else pad_token_id)
        terms["cap"] = d["cap_loss"]
    else:
        zero_keys += ["cap_loss", "cap_acc"]
    for k in zero_keys:
        d[k] = _const("zero", lambda: torch.zeros(()), dev)
    if fused_tail:   # one launch each way for the sum (and its backward, SelectBackward of the core's total included)
        loss, con = _LossTail.apply(core_out, terms["lang"], terms["lcon"], terms["icon"], terms["ans"], terms["cap"])
        if terms["lcon"] is not None:
            d["con_loss"] = con.detach()
    else:       # loss_joint.py:204-223, op by op
        if terms["lang"] is not None:
            loss = loss + 0.3 * terms["lang"]
        if terms["lcon"] is not None:
            loss = loss + d["con_loss"]
        if terms["ans"] is not None:
            loss = loss + terms["ans"]
        if terms["cap"] is not None:
            loss = loss + terms["cap"]
    d["loss"] = loss
    return d

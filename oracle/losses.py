"""ORACLE — TEST INFRASTRUCTURE ONLY.  Literal numpy restatement of the reference's training loss for the grounding
path, loops and all, in float64:

  compute_vote_loss              lib/loss_helper/loss_detection.py:24-72
  compute_objectness_loss        lib/loss_helper/loss_detection.py:74-113
  recover_assigned_gt_bboxes     lib/loss_helper/loss_detection.py:150-210
  compute_box_loss               lib/loss_helper/loss_detection.py:116-147
  compute_box_and_sem_cls_loss   lib/loss_helper/loss_detection.py:214-258
  box3d_diou_batch_tensor        utils/box_util.py:488-529            (pinned: fixture `boxes`)
  SoftmaxRankingLoss             lib/loss_helper/loss.py:6-17          (pinned: fixture `ranking_loss`)
  compute_diou_loss              lib/loss_helper/loss_grounding.py:129-365 (use_reg_head / use_kl_loss / debug off)
  get_joint_loss                 lib/loss_helper/loss_joint.py:26-227  (detection + reference [+ diou] [+ contrast])

loss_detection.py / loss_grounding.py cannot be imported in this container (easydict, torch_scatter: SURVEY.md §8c),
so these functions are pinned only through the two importable pieces named above; the rest is "parity unpinned" and
follows the source statement by statement (the Python loop over (scene, sentence) is kept).
The language-classification term (loss_grounding.py:476-487) reads `lang_scores` of the language encoder, which is
out of scope; it is restated for completeness and used only when the caller supplies those tensors.
"""
import numpy as np

from . import oracle as orc

FAR_THRESHOLD = 0.3   # loss_detection.py:19-22
NEAR_THRESHOLD = 0.3
GT_VOTE_FACTOR = 3
OBJECTNESS_CLS_WEIGHTS = (0.2, 0.8)


def _f64(a):
    return np.asarray(a, np.float64)


def huber(error, delta):
    a = np.abs(error)
    q = np.minimum(a, delta)
    return 0.5 * q ** 2 + delta * (a - q)


def _log_softmax(x):
    x = x - x.max(axis=-1, keepdims=True)
    return x - np.log(np.exp(x).sum(axis=-1, keepdims=True))


def cross_entropy(scores, label, weight=None):
    """nn.CrossEntropyLoss(weight, reduction='none') on (..., C) scores."""
    ls = _log_softmax(_f64(scores))
    ce = -np.take_along_axis(ls, label[..., None], -1)[..., 0]
    return ce if weight is None else ce * np.asarray(weight, np.float64)[label]


def compute_vote_loss(d):
    B, S = d["seed_xyz"].shape[:2]
    seed_inds = np.asarray(d["seed_inds"]).astype(np.int64)
    mask = np.take_along_axis(_f64(d["vote_label_mask"]), seed_inds, 1)
    gt = np.take_along_axis(_f64(d["vote_label"]), seed_inds[..., None].repeat(9, -1), 1)
    gt = gt + np.tile(_f64(d["seed_xyz"]), (1, 1, 3))
    vote = _f64(d["vote_xyz"]).reshape(B * S, -1, 3)
    gt = gt.reshape(B * S, GT_VOTE_FACTOR, 3)
    dist = np.abs(vote[:, :, None, :] - gt[:, None, :, :]).sum(-1)  # nn_distance(..., l1=True)
    dist2 = dist.min(axis=1)                                        # (B*S, GT_VOTE_FACTOR)
    votes_dist = dist2.min(axis=1).reshape(B, S)
    return float((votes_dist * mask).sum() / (mask.sum() + 1e-6))


def compute_objectness_loss(d):
    agg = np.asarray(d["aggregated_vote_xyz"], np.float32)
    gt_center = np.asarray(d["center_label"], np.float32)[:, :, 0:3]
    dist1, ind1, _, _ = orc.nn_distance(agg, gt_center)             # fp32, first minimum, like the reference op
    euc = np.sqrt(dist1 + np.float32(1e-6))
    label = np.zeros(euc.shape, np.int64)
    mask = np.zeros(euc.shape, np.float64)
    label[euc < NEAR_THRESHOLD] = 1
    mask[euc < NEAR_THRESHOLD] = 1
    mask[euc > FAR_THRESHOLD] = 1
    ce = cross_entropy(d["objectness_scores"], label, OBJECTNESS_CLS_WEIGHTS)
    return float((ce * mask).sum() / (mask.sum() + 1e-6)), label, mask, ind1


def rotz_batch(t):
    """utils/box_util.py:410-429 (the TRANSPOSED rotation, applied to row vectors)."""
    out = np.zeros(t.shape + (3, 3))
    c, s = np.cos(t), np.sin(t)
    out[..., 0, 0], out[..., 0, 1], out[..., 1, 0], out[..., 1, 1], out[..., 2, 2] = c, -s, s, c, 1
    return out


def recover_assigned_gt_bboxes(d, config, assign):
    nh = config["num_heading_bin"]
    mean_size = _f64(config["mean_size_arr"])
    agg = _f64(d["aggregated_vote_xyz"]).copy()
    B, K = assign.shape
    gt_center = np.take_along_axis(_f64(d["center_label"])[:, :, 0:3], assign[..., None].repeat(3, -1), 1)
    hcl = np.take_along_axis(np.asarray(d["heading_class_label"]), assign, 1)
    hrl = np.take_along_axis(_f64(d["heading_residual_label"]), assign, 1)
    gt_heading = hcl * (2 * np.pi / nh) + hrl if nh != 1 else np.zeros((B, K))
    scl = np.take_along_axis(np.asarray(d["size_class_label"]), assign, 1)
    srl = np.take_along_axis(_f64(d["size_residual_label"]), assign[..., None].repeat(3, -1), 1)
    gt_size = mean_size[scl] + srl
    half = gt_size / 2
    agg -= gt_center
    R = rotz_batch(-gt_heading)
    agg = np.einsum("bkc,bkcd->bkd", agg, R)                         # row vector @ R
    bld, fru = half + agg, half - agg
    return gt_center, hcl, hrl, gt_heading, np.concatenate([bld, fru], 2), gt_size


def compute_box_and_sem_cls_loss(d, config, label, assign):
    nh = config["num_heading_bin"]
    _, hcl, hrl, _, gt_distance, _ = recover_assigned_gt_bboxes(d, config, assign)
    lab = label.astype(np.float64)
    den = lab.sum() + 1e-6
    hc = float((cross_entropy(d["heading_scores"], hcl) * lab).sum() / den)
    onehot = np.zeros(hcl.shape + (nh,))
    np.put_along_axis(onehot, hcl[..., None], 1, -1)
    res = (_f64(d["heading_residuals_normalized"]) * onehot).sum(-1) - hrl / (np.pi / nh)
    hr = float((huber(res, 1.0) * lab).sum() / den)
    dist = huber(_f64(d["rois"]) - gt_distance, 0.15).mean(-1)
    dl = float((dist * lab).sum() / den)
    sem_label = np.take_along_axis(np.asarray(d["sem_cls_label"]), assign, 1)
    sem = float((cross_entropy(d["sem_cls_scores"], sem_label) * lab).sum() / den)
    return hc, hr, dl, sem


def box3d_diou(center1, size1, center2, size2):
    """utils/box_util.py:488-529 on (n,3) arrays -> (iou, diou)."""
    c1, s1, c2, s2 = (_f64(t) for t in (center1, size1, center2, size2))
    area1, area2 = s1.prod(-1), s2.prod(-1)
    lo, hi = np.maximum(c1 - s1 / 2, c2 - s2 / 2), np.minimum(c1 + s1 / 2, c2 + s2 / 2)
    inter = np.clip(hi - lo, 0, None).prod(-1)
    iou = inter / (area1 + area2 - inter)
    inter_diag = ((c1 - c2) ** 2).sum(-1)
    olo, ohi = np.minimum(c1 - s1 / 2, c2 - s2 / 2), np.maximum(c1 + s1 / 2, c2 + s2 / 2)
    outer_diag = (np.clip(ohi - olo, 0, None) ** 2).sum(-1)
    return iou, np.clip(iou - 1.5 * inter_diag / outer_diag, -1, 1)


def softmax_ranking_loss(inputs, targets):
    """lib/loss_helper/loss.py:6-17."""
    x = _f64(inputs) + 1e-8
    p = np.exp(_log_softmax(x))
    return float((-(np.log(p + 1e-8) * _f64(targets)).sum(-1)).mean())


def compute_diou_loss(d, config, no_reference=False):
    """loss_grounding.py:129-365 -> dict(ref_loss, diou_loss, cluster_labels (B,L,K), smooth_labels, rates)."""
    pred_center, pred_size = _f64(d["pred_center"]), _f64(d["pred_size"])
    gt_center_list = _f64(d["ref_center_label_list"])
    B, K = pred_center.shape[:2]
    L = gt_center_list.shape[1]
    lang_num = np.asarray(d["lang_num"]).astype(np.int64)
    mean_size = _f64(config["mean_size_arr"])
    cluster_preds = _f64(d["cluster_ref"]).reshape(B, L, K)
    obj_mask_all = (np.argmax(np.asarray(d["objectness_scores"]), 2)).astype(np.float64)
    gate = bool(d["istrain"][0] == 1 and not no_reference and float(d["random"]) < 0.5)
    loss = diou_loss = 0.0
    rate25 = rate5 = 0
    gt_labels = np.zeros((B, L, K))
    smooth_all = np.zeros((B, L, K))
    for i in range(B):
        gt_box_center = gt_center_list[i][:, 0:3]
        gt_box_size = mean_size[np.asarray(d["ref_size_class_label_list"])[i]] + _f64(d["ref_size_residual_label_list"])[i]
        labels = np.zeros((L, K))
        smooth = np.zeros((L, K))
        dious_rows = []
        for j in range(L):
            if j < lang_num[i]:
                ious, dious = box3d_diou(pred_center[i], pred_size[i], np.tile(gt_box_center[j], (K, 1)),
                                         np.tile(gt_box_size[j], (K, 1)))
                ious_np = ious.copy()
                dious_rows.append(dious)
                if gate:
                    ious = ious * obj_mask_all[i]
                ind = int(ious_np.argmax())
                if ious_np[ind] >= 0.25:
                    labels[j, ind] = 1
                    if d["epoch"] < 50:
                        sm = ious >= 0.25
                        cnt = int(sm.sum())
                        if cnt >= 2:
                            smooth[j, sm] = 0.05 / (cnt - 1)
                            smooth[j, int(ious.argmax())] = 0.95
                        else:
                            smooth[j, int(ious.argmax())] = 1
                    else:
                        smooth[j, int(ious.argmax())] = 1
                    rate25 += 1
                if ious_np[ind] >= 0.5:
                    rate5 += 1
        n = int(lang_num[i])
        gt_labels[i], smooth_all[i] = labels, smooth
        loss += softmax_ranking_loss(cluster_preds[i, :n], smooth[:n])
        diou_loss += float(((1 - np.stack(dious_rows)[:n]) * smooth[:n]).sum())
    tot = max(int(lang_num.sum()), 1)
    return dict(ref_loss=loss / B, diou_loss=diou_loss / B, cluster_labels=gt_labels, smooth_labels=smooth_all,
                max_iou_rate_25=rate25 / tot, max_iou_rate_5=rate5 / tot)


def compute_lang_classification_loss(d):
    """loss_grounding.py:476-487."""
    cats = np.asarray(d["object_cat_list"])
    B, L = cats.shape[:2]
    scores = _f64(d["lang_scores"]).reshape(B, L, -1)
    loss = 0.0
    for i in range(B):
        n = int(d["lang_num"][i])
        loss += float(cross_entropy(scores[i, :n], cats[i, :n]).mean())
    return loss / B


def get_joint_loss(d, config, use_diou_loss=True, use_con=True, use_lang_classifier=False):
    """loss_joint.py:26-227 with detection=True, reference=True, caption=False (run.sh:1 adds --use_con
    --use_diou_loss).  `d` additionally holds lang_con_loss / iou_con_loss when the contrast module ran."""
    out = {}
    out["vote_loss"] = compute_vote_loss(d)
    out["objectness_loss"], label, mask, assign = compute_objectness_loss(d)
    total = label.size
    out["pos_ratio"] = float(label.sum() / total)
    out["neg_ratio"] = float(mask.sum() / total - out["pos_ratio"])
    hc, hr, dl, sem = compute_box_and_sem_cls_loss(d, config, label, assign)
    out.update(heading_cls_loss=hc, heading_reg_loss=hr, size_distance_loss=dl, sem_cls_loss=sem)
    out["box_loss"] = 0.1 * hc + hr + 0.1 * sem + 20 * dl
    pred = np.argmax(np.asarray(d["objectness_scores"]), 2)
    out["obj_acc"] = float(((pred == label) * mask).sum() / (mask.sum() + 1e-6))
    r = compute_diou_loss(d, config)
    out.update(ref_loss=r["ref_loss"], diou_loss=r["diou_loss"], cluster_labels=r["cluster_labels"],
               smooth_labels=r["smooth_labels"], max_iou_rate_25=r["max_iou_rate_25"], max_iou_rate_5=r["max_iou_rate_5"],
               objectness_label=label, objectness_mask=mask, object_assignment=assign)
    loss = 10 * (out["vote_loss"] + 0.1 * out["objectness_loss"] + out["box_loss"])
    loss += (0.3 if d["epoch"] < 50 else 1.0) * out["ref_loss"]
    if use_diou_loss:
        loss += 0.3 * out["diou_loss"]
    if use_lang_classifier:
        out["lang_loss"] = compute_lang_classification_loss(d)
        loss += 0.3 * out["lang_loss"]
    if use_con and d["epoch"] >= 50:
        out["con_loss"] = 0.5 * float(d["lang_con_loss"]) + 2.5 * float(d["iou_con_loss"])
        loss += out["con_loss"]
    out["loss"] = loss
    return out


def answer_classification_loss(answer_scores, answer_cat_scores=None, answer_cat=None):
    """lib/loss_helper/loss_answering.py:2-16.  Soft scores: sum over (question, answer) of the BCE-with-logits
    max(x, 0) - x t + log(1 + exp(-|x|)) divided by the number of questions; class indices: mean cross entropy."""
    x = np.asarray(answer_scores, np.float64)
    if answer_cat_scores is not None:
        t = np.asarray(answer_cat_scores, np.float64)
        return float((np.maximum(x, 0) - x * t + np.log1p(np.exp(-np.abs(x)))).sum() / x.shape[0])
    m = x.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(x - m).sum(1))
    return float((lse - x[np.arange(x.shape[0]), np.asarray(answer_cat)]).mean())

"""Generate golden fixtures by IMPORTING the reference's own Python modules.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The fixtures (tests/golden/*.npz) are DATA: inputs, parameter values and the
reference's outputs.  No reference source text is stored.  Parameters and inputs
are drawn on an int8 grid (value = int8 / scale) so that they are exactly
representable and the files stay small.

Pinned here (SURVEY.md §8c list): nn_distance; ScaledDotProductAttention /
MultiHeadAttention (plain, additive bias, multiplicative weights, mask);
CrossAttentionDecoderLayer x2; MatchModule (eval, istrain=0); RelationModule
(eval); VotingModule (train+eval); SharedMLP (train+eval);
box3d_diou_batch_tensor; StandardROIHeads (eval); get_3d_box_batch; SoftmaxRankingLoss; AnswerModule (eval).

NOT pinnable (documented in DESIGN.md): the nine pointnet2._ext CUDA ops and
pytorch3d box3d_overlap.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path[:0] = [REF, os.path.join(REF, "lib", "pointnet2")]
sys.dont_write_bytecode = True

from utils.nn_distance import nn_distance  # noqa: E402
from utils.box_util import box3d_diou_batch_tensor, get_3d_box_batch  # noqa: E402
from models.transformer.attention import MultiHeadAttention, ScaledDotProductAttention  # noqa: E402
from models.transformer.mmattention import CrossAttentionDecoderLayer  # noqa: E402
from models.base_module.voting_module import VotingModule  # noqa: E402
from models.refnet.match_module import MatchModule  # noqa: E402
from models.proposal_module.relation_module import RelationModule  # noqa: E402
from models.proposal_module.ROI_heads.roi_heads import StandardROIHeads  # noqa: E402
import pytorch_utils  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def qrand(rng, shape, scale, lo=-48, hi=48):
    """int8-grid random tensor: returns (int8 array, float32 value = q/scale)."""
    q = rng.integers(lo, hi + 1, size=shape, dtype=np.int64).astype(np.int8)
    return q, (q.astype(np.float32) / np.float32(scale))


def quantize_module_(module, rng):
    """Overwrite every parameter / float buffer with int8-grid values sized by fan-in.
    Returns {name: (int8 array, scale)}."""
    store = {}
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            if not t.dtype.is_floating_point:
                continue
            shape = tuple(t.shape)
            leaf = name.split(".")[-1]
            if leaf == "running_var":
                q = rng.integers(32, 96, size=shape).astype(np.int8)
                scale = 64.0
            elif leaf in ("running_mean", "bias") or t.dim() <= 1:
                q = rng.integers(-16, 17, size=shape).astype(np.int8)
                scale = 64.0
                if leaf == "weight":  # norm gains around 1
                    q = rng.integers(48, 80, size=shape).astype(np.int8)
            else:
                fan_in = int(np.prod(shape[1:]))
                q = rng.integers(-32, 33, size=shape).astype(np.int8)
                scale = float(2 ** int(np.ceil(np.log2(32 * np.sqrt(fan_in / 3.0)))))
            val = torch.from_numpy(np.asarray(q.astype(np.float32) / np.float32(scale), np.float32)).reshape(shape)
            t.copy_(val)
            store[name] = (q, scale)
    return store


def pack(store, prefix="w/"):
    d = {}
    for k, (q, s) in store.items():
        d[prefix + k] = q
        d[prefix + k + "#scale"] = np.float32(s)
    return d


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def t2n(t):
    return t.detach().cpu().numpy()


def gen_nn_distance():
    d = {}
    for seed, (n, m, b) in enumerate([(5, 6, 1), (5, 6, 1), (256, 64, 4), (33, 129, 2)]):
        rng = np.random.default_rng(seed)
        pc1 = rng.random((b, n, 3)).astype(np.float32)
        pc2 = rng.random((b, m, 3)).astype(np.float32)
        if seed == 0:  # the reference demo's inputs (utils/nn_distance.py:95-101)
            np.random.seed(0)
            pc1 = np.random.random((1, 5, 3)).astype(np.float32)
            pc2 = np.random.random((1, 6, 3)).astype(np.float32)
        d[f"{seed}/pc1"], d[f"{seed}/pc2"] = pc1, pc2
        for tag, kw in (("l2", {}), ("l1", {"l1": True}), ("huber", {"l1smooth": True, "delta": 0.5})):
            o = nn_distance(torch.from_numpy(pc1), torch.from_numpy(pc2), **kw)
            for nm, t in zip(("dist1", "idx1", "dist2", "idx2"), o):
                d[f"{seed}/{tag}/{nm}"] = t2n(t)
    save("nn_distance", **d)


def gen_attention():
    rng = np.random.default_rng(100)
    torch.manual_seed(0)
    B, nq, nk, dm, h = 2, 48, 49, 128, 4
    mha = MultiHeadAttention(d_model=dm, d_k=dm // h, d_v=dm // h, h=h).eval()
    store = quantize_module_(mha, rng)
    d = pack(store)
    qq, q = qrand(rng, (B, nq, dm), 32)
    kq, k = qrand(rng, (B, nk, dm), 32)
    bq, bias = qrand(rng, (B, h, nq, nk), 16)
    wq, wts = qrand(rng, (B, h, nq, nk), 32, 0, 64)
    mask = (rng.random((B, 1, 1, nk)) > 0.3).astype(np.float32)
    mask[..., 0] = 1
    d.update({"in/q": qq, "in/q#scale": np.float32(32), "in/k": kq, "in/k#scale": np.float32(32),
              "in/bias": bq, "in/bias#scale": np.float32(16), "in/wts": wq, "in/wts#scale": np.float32(32),
              "in/mask": mask})
    Q, K = torch.from_numpy(q), torch.from_numpy(k)
    with torch.no_grad():
        o, a = mha(Q, K, K, output_attn=True)
        d["out/cross"], d["out/cross_att"] = t2n(o), t2n(a)
        o, a = mha(Q, Q, Q, output_attn=True)
        d["out/self"] = t2n(o)
        o, a = mha(Q, K, K, attention_weights=torch.from_numpy(bias), way="add", output_attn=True)
        d["out/add"], d["out/add_att"] = t2n(o), t2n(a)
        o, a = mha(Q, K, K, attention_weights=torch.from_numpy(wts), way="mul", output_attn=True)
        d["out/mul"] = t2n(o)
        o, a = mha(Q, K, K, attention_mask=torch.from_numpy(mask), output_attn=True)
        d["out/mask"] = t2n(o)
        o, a = mha.attention(Q, K, K)
        d["out/sdpa"] = t2n(o)
    save("attention", **d)


def gen_decoder():
    rng = np.random.default_rng(200)
    B, nq, nk, dm = 2, 40, 49, 128
    layers = torch.nn.ModuleList(CrossAttentionDecoderLayer(hidden_size=dm) for _ in range(2)).eval()
    store = quantize_module_(layers, rng)
    d = pack(store)
    qq, q = qrand(rng, (B, nq, dm), 32)
    kq, k = qrand(rng, (B, nk, dm), 32)
    d.update({"in/q": qq, "in/q#scale": np.float32(32), "in/k": kq, "in/k#scale": np.float32(32)})
    x = torch.from_numpy(q)
    K = torch.from_numpy(k)
    with torch.no_grad():
        for i in range(2):
            x = layers[i](x, K, K)
            d[f"out/layer{i}"] = t2n(x)
    save("decoder_layer", **d)


def gen_match():
    rng = np.random.default_rng(300)
    B, P, L, dm = 2, 32, 2, 128
    m = MatchModule(num_proposals=P, lang_size=256, det_channel=128).eval()
    store = quantize_module_(m, rng)
    d = pack(store)
    fq, feat = qrand(rng, (B, P, dm), 32)
    lq, lang = qrand(rng, (B * L, 50, dm), 32)
    oq, obj = qrand(rng, (B, P, 2), 16)
    d.update({"in/bbox_feature": fq, "in/bbox_feature#scale": np.float32(32),
              "in/lang_fea": lq, "in/lang_fea#scale": np.float32(32),
              "in/objectness_scores": oq, "in/objectness_scores#scale": np.float32(16)})
    dd = {"objectness_scores": torch.from_numpy(obj), "bbox_feature": torch.from_numpy(feat),
          "input_ids": torch.zeros(B, L, 50, dtype=torch.long), "istrain": [0],
          "lang_fea": torch.from_numpy(lang)}
    with torch.no_grad():
        dd = m(dd)
    d["out/cluster_ref"] = t2n(dd["cluster_ref"])
    d["out/cross_box_feature"] = t2n(dd["cross_box_feature"])
    save("match_module", **d)


def gen_relation():
    rng = np.random.default_rng(400)
    B, P, N, S = 2, 32, 256, 64
    m = RelationModule(num_proposals=P, det_channel=128).eval()
    store = quantize_module_(m, rng)
    d = pack(store)
    fq, feat = qrand(rng, (B, P, 128), 32)
    cq, corner = qrand(rng, (B, P, 8, 3), 16)
    pq, pc = qrand(rng, (B, N, 135), 32)
    seed_inds = rng.integers(0, N, size=(B, S)).astype(np.int32)
    vote_inds = rng.integers(0, S, size=(B, P)).astype(np.int32)
    d.update({"in/pred_bbox_feature": fq, "in/pred_bbox_feature#scale": np.float32(32),
              "in/pred_bbox_corner": cq, "in/pred_bbox_corner#scale": np.float32(16),
              "in/point_clouds": pq, "in/point_clouds#scale": np.float32(32),
              "in/seed_inds": seed_inds, "in/aggregated_vote_inds": vote_inds})
    dd = {"pred_bbox_feature": torch.from_numpy(feat), "pred_bbox_corner": torch.from_numpy(corner),
          "point_clouds": torch.from_numpy(pc), "seed_inds": torch.from_numpy(seed_inds),
          "aggregated_vote_inds": torch.from_numpy(vote_inds)}
    with torch.no_grad():
        dd = m(dd)
    d["out/bbox_feature"] = t2n(dd["bbox_feature"])
    d["out/dist_weights"] = t2n(dd["dist_weights"])
    save("relation_module", **d)


def gen_voting():
    rng = np.random.default_rng(500)
    B, S = 2, 64
    m = VotingModule(1, 256)
    store = quantize_module_(m, rng)
    d = pack(store)
    xq, xyz = qrand(rng, (B, S, 3), 16)
    fq, feat = qrand(rng, (B, 256, S), 32)
    d.update({"in/seed_xyz": xq, "in/seed_xyz#scale": np.float32(16),
              "in/seed_features": fq, "in/seed_features#scale": np.float32(32)})
    with torch.no_grad():
        m.eval()
        vx, vf = m(torch.from_numpy(xyz), torch.from_numpy(feat))
        d["out/eval/vote_xyz"], d["out/eval/vote_features"] = t2n(vx), t2n(vf)
        m.train()
        vx, vf = m(torch.from_numpy(xyz), torch.from_numpy(feat))
        d["out/train/vote_xyz"], d["out/train/vote_features"] = t2n(vx), t2n(vf)
    save("voting_module", **d)


def gen_shared_mlp():
    rng = np.random.default_rng(600)
    m = pytorch_utils.SharedMLP([135, 64, 64, 128], bn=True)
    store = quantize_module_(m, rng)
    d = pack(store)
    d["meta/keys"] = np.array(sorted(m.state_dict().keys()))
    xq, x = qrand(rng, (2, 135, 32, 16), 32)
    d.update({"in/x": xq, "in/x#scale": np.float32(32)})
    with torch.no_grad():
        m.eval()
        d["out/eval"] = t2n(m(torch.from_numpy(x)))
        m.train()
        d["out/train"] = t2n(m(torch.from_numpy(x)))
    save("shared_mlp", **d)


def gen_boxes():
    rng = np.random.default_rng(700)
    n = 256
    c1 = rng.uniform(0, 4, (n, 3)).astype(np.float32)
    s1 = rng.uniform(0.3, 2, (n, 3)).astype(np.float32)
    c2 = (c1 + rng.normal(0, 0.5, (n, 3))).astype(np.float32)
    s2 = rng.uniform(0.3, 2, (n, 3)).astype(np.float32)
    iou, diou = box3d_diou_batch_tensor(*(torch.from_numpy(a) for a in (c1, s1, c2, s2)))
    heading = rng.uniform(-3, 3, (4, 16)).astype(np.float32)
    size = rng.uniform(0.3, 2, (4, 16, 3)).astype(np.float32)
    center = rng.uniform(0, 4, (4, 16, 3)).astype(np.float32)
    corners = get_3d_box_batch(size, heading, center)
    save("boxes", c1=c1, s1=s1, c2=c2, s2=s2, iou=t2n(iou), diou=t2n(diou), heading=heading, size=size,
         center=center, corners=corners.astype(np.float32))


def gen_roi_heads():
    rng = np.random.default_rng(800)
    B, P = 2, 32
    m = StandardROIHeads(num_heading_bin=1, num_class=18, seed_feat_dim=256).eval()
    store = quantize_module_(m, rng)
    d = pack(store)
    fq, feat = qrand(rng, (B, 128, P), 32)
    d.update({"in/features": fq, "in/features#scale": np.float32(32)})
    with torch.no_grad():
        dd = m(torch.from_numpy(feat), {})
    for k in ("objectness_scores", "rois", "heading_scores", "heading_residuals_normalized", "heading_residuals",
              "sem_cls_scores"):
        d["out/" + k] = t2n(dd[k])
    save("roi_heads", **d)


def gen_ranking_loss():
    """lib/loss_helper/loss.py:6-17 SoftmaxRankingLoss (pure torch, importable): hard, smooth (0.95 / 0.05 split) and
    all-zero target rows, as compute_diou_loss feeds it (loss_grounding.py:258-300)."""
    from lib.loss_helper.loss import SoftmaxRankingLoss
    rng = np.random.default_rng(900)
    crit = SoftmaxRankingLoss()
    d = {}
    for case, (rows, K) in enumerate([(8, 256), (3, 32), (1, 5)]):
        x = (rng.normal(0, 3, (rows, K))).astype(np.float32)
        t = np.zeros((rows, K), np.float32)
        for r in range(rows):
            kind = r % 3
            if kind == 0:
                t[r, rng.integers(0, K)] = 1
            elif kind == 1:
                sel = rng.choice(K, size=min(4, K), replace=False)
                t[r, sel] = 0.05 / (len(sel) - 1)
                t[r, sel[0]] = 0.95
        d[f"{case}/x"], d[f"{case}/t"] = x, t
        d[f"{case}/loss"] = t2n(crit(torch.from_numpy(x), torch.from_numpy(t)))
    save("ranking_loss", **d)


def gen_answer():
    """models/answer_module/answer_module.py:10-114 (QA head as shipped: AttFlat over cross_box_feature -> answer_cls), eval."""
    from models.answer_module.answer_module import AnswerModule
    rng = np.random.default_rng(1000)
    m = AnswerModule(num_answers=24, hidden_size=128).eval()
    d = pack(quantize_module_(m, rng))
    xq, x = qrand(rng, (6, 32, 128), 32)
    d.update({"in/cross_box_feature": xq, "in/cross_box_feature#scale": np.float32(32)})
    with torch.no_grad():
        out = m({"cross_box_feature": torch.from_numpy(x)})
    d["out/answer_scores"] = t2n(out["answer_scores"])
    save("answer_module", **d)


if __name__ == "__main__":
    if "--only-ranking" in sys.argv:
        gen_ranking_loss()
        sys.exit(0)
    if "--only-answer" in sys.argv:
        gen_answer()
        sys.exit(0)
    gen_nn_distance()
    gen_attention()
    gen_decoder()
    gen_match()
    gen_relation()
    gen_voting()
    gen_shared_mlp()
    gen_boxes()
    gen_roi_heads()
    gen_ranking_loss()
    gen_answer()

"""Language-grounding half of the path: proposal<->token matching and the OCC/OSC contrastive losses.

  MatchModule    — models/refnet/match_module.py:10-170 (2x CrossAttentionDecoderLayer + match MLP)
  ContrastModule — models/constrast_module/constrast_module.py:9-131 (NCELoss :24-37)
State-dict keys follow the reference (unused sub-modules such as lang_emb_proj / box_con_proj are kept
for checkpoint compatibility).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as _ext
from . import add_norm, glue, row_chain
from .mfma_linear import linear as _linear
from .transformer import CrossAttentionDecoderLayer, MultiHeadAttention, decoder_stack_chained


class MatchModule(nn.Module):
    def __init__(self, num_proposals=256, lang_size=256, hidden_size=128, lang_num_size=300, det_channel=128, head=4,
                 use_lang_emb=False, use_pc_encoder=False, use_match_con_loss=False, depth=2, use_reg_head=False):
        super().__init__()
        self.num_proposals, self.lang_size, self.hidden_size = num_proposals, lang_size, hidden_size
        self.use_lang_emb, self.use_pc_encoder, self.depth = use_lang_emb, use_pc_encoder, depth
        self.use_reg_head = use_reg_head
        self.match = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.GELU(), nn.Dropout(p=0.5, inplace=False),
                                   nn.Linear(hidden_size, hidden_size), nn.GELU(), nn.Dropout(p=0.5, inplace=False),
                                   nn.Linear(hidden_size, 1))
        if self.use_reg_head:
            self.reg_head = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.BatchNorm1d(hidden_size), nn.GELU(),
                                          nn.Linear(hidden_size, hidden_size), nn.BatchNorm1d(hidden_size), nn.GELU(),
                                          nn.Linear(hidden_size, 6), nn.Sigmoid())
        self.lang_emb_proj = nn.Sequential(nn.Conv1d(hidden_size, hidden_size, 1), nn.BatchNorm1d(hidden_size),
                                           nn.PReLU(), nn.Conv1d(hidden_size, hidden_size, 1),
                                           nn.BatchNorm1d(hidden_size), nn.PReLU(),
                                           nn.Conv1d(hidden_size, num_proposals, 1))
        self.grounding_cross_attn = nn.ModuleList(
            CrossAttentionDecoderLayer(hidden_size=hidden_size) for _ in range(self.depth))
        self.lang_emb_cross_attn = MultiHeadAttention(d_model=hidden_size, d_k=hidden_size // head,
                                                      d_v=hidden_size // head, h=head)
        self.loss_fn = nn.CrossEntropyLoss()
        self.box_con_proj = nn.Linear(hidden_size, hidden_size)
        self.lang_con_proj = nn.Linear(hidden_size, hidden_size)
        self.temp = nn.Parameter(torch.ones([]) * 0.07)
        self.use_match_con_loss = use_match_con_loss

    def unused_batchnorms(self):
        """BatchNorm layers that exist for checkpoint compatibility but never run in this configuration: a step driver
        that increments all `num_batches_tracked` in one launch (grounding_step._deferred_bn_counters) must skip them —
        the reference's counters only advance for layers whose forward ran."""
        if self.use_lang_emb:
            return []
        return [m for m in self.lang_emb_proj.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]

    @staticmethod
    def _copy_paste(features, objectness_masks):
        """Train-time augmentation (:97-121): background proposals of scene i are overwritten with object
        features taken from the (twice repeated) batch-wide list of object proposals, starting right after
        scene i's own objects.  The reference drives this from the host (torch.where + Python slicing, one
        sync per scene); here the same assignment is a fixed-shape gather, so the step stays capturable in a
        HIP graph:  background slot of rank r in scene i  <-  pool[(J_i + r) mod total]  if r < total - n_i,
        with J_i = objects in scenes 0..i, pool = object proposals in (scene, proposal) order."""
        B, K, D = features.shape
        obj = objectness_masks.bool().squeeze(2)            # (B,K)
        n_obj = obj.sum(1)                                  # (B,)
        total = n_obj.sum()
        J = torch.cumsum(n_obj, 0)                          # (B,)
        bg = ~obj
        rank = torch.cumsum(bg.long(), 1) - 1               # rank of each background slot inside its scene
        take = bg & (rank < (total - n_obj)[:, None])
        obj_pos = torch.argsort((~obj.reshape(-1)).to(torch.int8), stable=True)  # object slots first, in order
        src = obj_pos[(J[:, None] + rank).clamp(min=0) % total.clamp(min=1)]     # (B,K) flat source slot
        pasted = torch.gather(features.reshape(B * K, D), 0, src.reshape(-1, 1).expand(-1, D)).reshape(B, K, D)
        return torch.where(take.unsqueeze(-1), pasted, features)

    def forward(self, data_dict):
        features = data_dict["bbox_feature"]  # (B, K, hidden)
        B, K = features.shape[:2]
        L = data_dict["input_ids"].shape[1]
        feature0 = features
        if data_dict["istrain"][0] == 1:
            # the reference draws random.random() < 0.5 on the host; a device-side draw keeps the step free of
            # host decisions (graph-capturable) with the same 50 % rate
            coin = data_dict.get("random")
            if coin is None:
                coin = torch.rand((), device=features.device)
            data_dict["random"] = coin
            coin = torch.as_tensor(coin, device=features.device)
            obj_mask = data_dict["bbox_mask"] if "bbox_mask" in data_dict else data_dict["objectness_scores"].max(2)[1]
            if features.is_cuda and features.dtype == torch.float32 and B * K <= 8192 and features.shape[-1] % 4 == 0:
                feature0 = glue.copy_paste(features, obj_mask, coin)  # index map + row gather: two launches
            else:
                feature0 = torch.where(coin < 0.5, self._copy_paste(features, obj_mask.float().unsqueeze(2)), features)

        # K/V = the tokens after [CLS]; the loader may hand over the contiguous copy (grounding_step.batch_to_device)
        lang_fea = data_dict["k/lang_kv"] if "k/lang_kv" in data_dict else data_dict["lang_fea"][:, 1:]

        # the proposals are tiled over the L sentences of their scene (:127); the first layer takes the UN-tiled features and
        # tiles after its (copy-independent) self-attention block — same values, see CrossAttentionDecoderLayer.forward_tiled
        layers = list(self.grounding_cross_attn)
        mods, i = list(self.match), 0
        # the row-local runs between the attention cores as one launch each (transformer.decoder_stack_chained), the first two
        # Linear/GELU/Dropout pairs of the match MLP riding on the last one; None: a shape / mode the chain kernel does not cover
        tail = []
        while (len(mods) >= i + 3 and isinstance(mods[i], nn.Linear) and isinstance(mods[i + 1], nn.GELU)
               and mods[i + 1].approximate == "none" and isinstance(mods[i + 2], nn.Dropout) and len(tail) < 2):
            tail.append(row_chain.linear(mods[i].weight, mods[i].bias, "gelu", mods[i + 2].p))
            i += 3
        chained = decoder_stack_chained(layers, feature0.contiguous(), L, lang_fea, tail)
        if chained is not None:
            feature1, x = chained
            if x is None:
                x, i = feature1.reshape(B * L * K, -1), 0
        else:
            feature1 = layers[0].forward_tiled(feature0.contiguous(), L, lang_fea, lang_fea)  # (B*L, K, hidden)
            for layer in layers[1:]:
                feature1 = layer(feature1, lang_fea, lang_fea)
            x, i = feature1.reshape(B * L * K, -1), 0
        data_dict["cross_box_feature"] = feature1

        feature1_agg = feature1.reshape(B * L * K, -1)
        while i < len(mods):  # nn.Sequential of Linear / GELU / Dropout: Linears on the MFMA kernels, GELU+Dropout fused
            layer = mods[i]
            if isinstance(layer, nn.Linear) and glue.rowdot_supported(x, layer.weight) and not torch.is_autocast_enabled("cuda"):
                x = glue.rowdot(x, layer.weight, layer.bias)  # Linear(128, 1): a row dot product
            elif isinstance(layer, nn.Linear):
                x = _linear(x, layer.weight, layer.bias)
            elif (isinstance(layer, nn.GELU) and layer.approximate == "none" and i + 1 < len(mods)
                  and isinstance(mods[i + 1], nn.Dropout) and add_norm.act_dropout_supported(x)):
                x = add_norm.act_dropout(x, "gelu", mods[i + 1].p, self.training)
                i += 1
            else:
                x = layer(x)
            i += 1
        confidence = x.squeeze(1).view(B * L, K)

        if self.use_lang_emb:
            lang_emb = data_dict["lang_emb"]
            lang_num_max = lang_emb.shape[0] // B
            lang_emb_feature = self.lang_emb_cross_attn(lang_emb.view(B, lang_num_max, -1), feature0, feature0)
            lang_emb_feature = lang_emb_feature.view(B * lang_num_max, -1, 1).contiguous()
            confidence = confidence + self.lang_emb_proj(lang_emb_feature).squeeze(2)
        data_dict["cluster_ref"] = confidence

        if self.use_reg_head:
            box_reg = (self.reg_head(feature1_agg) * 0.1 - 0.05).view(B, L, K, 6)
            data_dict["pred_center_reg"], data_dict["pred_size_reg"] = box_reg[..., 0:3], box_reg[..., 3:6]
        return data_dict


def axis_aligned_iou(center1, size1, center2, size2):
    """Closed-form IoU of axis-aligned boxes, broadcasting over leading dims.  Stands in for
    pytorch3d.ops.box3d_overlap (constrast_module.py:105) whose inputs are always axis aligned
    (create_box_batch :9-15); value-identical to the `iou` of utils/box_util.py:488-529."""
    lo = torch.max(center1 - size1 / 2, center2 - size2 / 2)
    hi = torch.min(center1 + size1 / 2, center2 + size2 / 2)
    inter = torch.clamp(hi - lo, min=0).prod(-1)
    return inter / (size1.prod(-1) + size2.prod(-1) - inter)


class NCELoss(nn.Module):
    """Holds the (unused, :32-33 commented out) temperature so that checkpoints load."""

    def __init__(self, init_tau=0.07, clamp=4.6051):
        super().__init__()
        self.tau = nn.Parameter(torch.tensor([np.log(1.0 / init_tau)], dtype=torch.float32))
        self.clamp = clamp


class _ContrastCore(torch.autograd.Function):
    """Fused OCC/OSC core (csrc/contrast.hip): normalised (text, box, boxi) + boxes -> (lang_con_loss, iou_con_loss)."""

    @staticmethod
    def forward(ctx, text, box, boxi, obj, gt_center, gt_size, pred_center, pred_size, lang_num):
        text, box, boxi = text.contiguous().float(), box.contiguous().float(), boxi.contiguous().float()
        cf = lambda t: t.contiguous().float()
        obj, gt_center, gt_size, pred_center, pred_size = cf(obj), cf(gt_center), cf(gt_size), cf(pred_center), cf(pred_size)
        lang_num = lang_num.contiguous().to(torch.int64)
        B, L, D = text.shape
        K = box.shape[1]
        out = torch.empty((2,), dtype=torch.float32, device=text.device)
        lse = torch.empty((B, L + K), dtype=torch.float32, device=text.device)
        _ext.call("vlp3d_contrast_fwd", text, box, boxi, obj, gt_center, gt_size, pred_center, pred_size, lang_num,
                  B, L, K, D, out, lse)
        ctx.save_for_backward(text, box, boxi, obj, gt_center, gt_size, pred_center, pred_size, lang_num, lse)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_occ, g_osc):
        text, box, boxi, obj, gt_center, gt_size, pred_center, pred_size, lang_num, lse = ctx.saved_tensors
        B, L, D = text.shape
        K = box.shape[1]
        dev = text.device
        dS = torch.empty((B, L + K, K), dtype=torch.float32, device=dev)
        dtext, dbox, dboxi = torch.empty_like(text), torch.empty_like(box), torch.empty_like(boxi)
        opt = lambda g: None if g is None else g.contiguous().float()
        _ext.call("vlp3d_contrast_bwd", text, box, boxi, obj, gt_center, gt_size, pred_center, pred_size, lang_num,
                  B, L, K, D, lse, opt(g_occ), opt(g_osc), dS, dtext, dbox, dboxi)
        return dtext, dbox, dboxi, None, None, None, None, None, None


class ContrastModule(nn.Module):
    """OCC (sentence vs proposals) and OSC (proposal vs proposal) InfoNCE, all (scene, sentence) pairs
    at once instead of the reference's Python double loop with host syncs.

    Semantics copied literally (SURVEY.md §7 hard part 4): no temperature; SoftCrossEntropy is the MEAN
    over all elements of -log_softmax * target (:18-21); for OCC the (1,P) logits make the transposed
    term vanish, so OCC = loss_v / 2; for OSC the logits are symmetric so loss_t == loss_v; targets are
    hard IoU > 0.25 masks against the GT box grown by 1e-2; only proposals whose objectness argmax is 1
    take part; sums over sentences are divided by batch_size; a no-op before epoch 50.
    `config` supplies mean_size_arr (decode of the GT size, model_util_scannet.py:183-190).
    """

    def __init__(self, config, hidden=128):
        super().__init__()
        self.pc_proj = nn.Linear(hidden, hidden, bias=False)
        self.text_proj = nn.Linear(hidden, hidden, bias=False)
        self.nce_loss = NCELoss()
        self.config = config
        self.bce_loss = nn.BCEWithLogitsLoss()
        self.pc_proj_iou = nn.Sequential(nn.Linear(hidden, hidden, bias=False))
        self._mean_size = None  # device copy of config.mean_size_arr (not a parameter/buffer: keeps the state_dict)
        self.fused = True  # csrc/contrast.hip on CUDA tensors; False = the batched op-by-op form below (host tests)

    def forward(self, data_dict):
        if data_dict["epoch"] < 50:
            data_dict["con_loss"] = torch.zeros(1)
            return data_dict
        pred_center = data_dict["pred_center"].detach()
        pred_size = data_dict["pred_size"].detach()
        features = data_dict["bbox_feature"]  # (B,K,hidden)
        B, K = features.shape[:2]
        gt_center = data_dict["ref_center_label_list"].detach()[..., 0:3]  # (B,L,3)
        L = gt_center.shape[1]
        if self._mean_size is None or self._mean_size.device != features.device:
            self._mean_size = torch.as_tensor(self.config.mean_size_arr, dtype=torch.float32, device=features.device)
        mean_size = self._mean_size
        gt_size = data_dict["k/ref_size"] if "k/ref_size" in data_dict else \
            mean_size[data_dict["ref_size_class_label_list"]] + data_dict["ref_size_residual_label_list"]
        lang_emb = data_dict["lang_emb"].view(B, -1, data_dict["lang_emb"].shape[-1])[:, :L]
        obj = data_dict["objectness_scores"].max(2)[1].to(features.dtype)  # (B,K) 1 = takes part
        if self.fused and features.is_cuda and K <= 1024 and L <= 64 and features.shape[-1] % 4 == 0:
            # three launches (csrc/contrast.hip) instead of ~45 + ~45 in autograd's backward
            text = glue.l2norm_rows(_linear(lang_emb.contiguous(), self.text_proj.weight))
            box = glue.l2norm_rows(_linear(features, self.pc_proj.weight))
            boxi = glue.l2norm_rows(_linear(features, self.pc_proj_iou[0].weight))
            data_dict["lang_con_loss"], data_dict["iou_con_loss"] = _ContrastCore.apply(
                text, box, boxi, obj, gt_center, gt_size, pred_center, pred_size, data_dict["lang_num"])
            return data_dict
        P = obj.sum(1)  # (B,) proposals taking part
        lang_ok = (torch.arange(L, device=features.device)[None, :] < data_dict["lang_num"][:, None]).to(features.dtype)
        lang_ok = lang_ok * (P > 0).to(features.dtype)[:, None]  # the reference's try/except skips empty scenes

        ious = axis_aligned_iou(gt_center[:, :, None, :], gt_size[:, :, None, :] + 1e-2, pred_center[:, None, :, :],
                                pred_size[:, None, :, :])  # (B,L,K)
        target = (ious > 0.25).to(features.dtype) * obj[:, None, :]
        neg_inf = -1e30  # excluded columns; their log-probabilities are zeroed before use (0 * -inf = NaN)
        Psafe = P.clamp(min=1)

        # OCC: sim (B,L,K) over the participating proposals
        text = F.normalize(self.text_proj(lang_emb), dim=-1)
        box = F.normalize(self.pc_proj(features), dim=-1)
        sim = torch.einsum("bld,bkd->blk", text, box).masked_fill(obj[:, None, :] == 0, neg_inf)
        logp = F.log_softmax(sim, dim=-1).masked_fill(obj[:, None, :] == 0, 0.0)
        loss_v = -(logp * target).sum(-1) / Psafe[:, None]  # mean over the (1,P) row
        lang_con_loss = (0.5 * loss_v * lang_ok).sum() / B

        # OSC: sim (B,K,K) among participating proposals, target = outer(mask, mask)
        boxi = F.normalize(self.pc_proj_iou(features), dim=-1)
        simi = torch.einsum("bkd,bjd->bkj", boxi, boxi).masked_fill(obj[:, None, :] == 0, neg_inf)
        nls = -F.log_softmax(simi, dim=-1).masked_fill(obj[:, None, :] == 0, 0.0)  # excluded columns -> 0
        quad = torch.einsum("blk,bkj,blj->bl", target, nls, target) / (Psafe * Psafe)[:, None]
        iou_con_loss = (quad * lang_ok).sum() / B

        data_dict["lang_con_loss"] = lang_con_loss
        data_dict["iou_con_loss"] = iou_con_loss
        return data_dict

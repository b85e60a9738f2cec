"""QA head of the ScanQA task: `AnswerModule` as the reference ships it
(models/answer_module/answer_module.py:10-114): answer_scores = answer_cls(AttFlat(cross_box_feature)), with AttFlat / MLP / FC
of models/vqa/mcan_module.py:18-112.  Same attribute names and state-dict keys (the reference also constructs — and never
calls — lang_feat_linear, object_feat_linear, object_cls and attflat_lang; they are kept so that checkpoints load strictly).
The linear layers run on the MFMA kernels (mfma_linear), GELU + Dropout as one launch (add_norm.act_dropout).
The caption head of the joint model is 3dvlp_amd/caption.py."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import add_norm
from .mfma_linear import linear as _linear


class FC(nn.Module):
    """Linear -> GELU -> Dropout (mcan_module.py:18-43)."""

    def __init__(self, in_size, out_size, pdrop=0., use_gelu=True):
        super().__init__()
        self.pdrop, self.use_gelu = pdrop, use_gelu
        self.linear = nn.Linear(in_size, out_size)
        if use_gelu:
            self.gelu = nn.GELU()
        if pdrop > 0:
            self.dropout = nn.Dropout(pdrop)

    def forward(self, x):
        x = _linear(x, self.linear.weight, self.linear.bias)
        if self.use_gelu and add_norm.act_dropout_supported(x) and not torch.is_autocast_enabled("cuda"):
            return add_norm.act_dropout(x, "gelu", self.pdrop, self.training)
        if self.use_gelu:
            x = self.gelu(x)
        if self.pdrop > 0:
            x = self.dropout(x)
        return x


class MLP(nn.Module):
    """FC -> Linear (mcan_module.py:46-54)."""

    def __init__(self, in_size, mid_size, out_size, pdrop=0., use_gelu=True):
        super().__init__()
        self.fc = FC(in_size, mid_size, pdrop=pdrop, use_gelu=use_gelu)
        self.linear = nn.Linear(mid_size, out_size)

    def forward(self, x):
        return _linear(self.fc(x), self.linear.weight, self.linear.bias)  # out_size = glimpses (1): library GEMV, counted as a fall-back


class AttFlat(nn.Module):
    """Attention pooling over the sequence (mcan_module.py:74-112): softmax over tokens of an MLP score per glimpse,
    weighted sums of x, concatenated and merged."""

    def __init__(self, hidden_size, flat_mlp_size=512, flat_glimpses=1, flat_out_size=1024, pdrop=0.1):
        super().__init__()
        self.mlp = MLP(in_size=hidden_size, mid_size=flat_mlp_size, out_size=flat_glimpses, pdrop=pdrop, use_gelu=True)
        self.flat_glimpses = flat_glimpses
        self.linear_merge = nn.Linear(hidden_size * flat_glimpses, flat_out_size)

    def forward(self, x, x_mask):
        att = self.mlp(x)  # (b, n, glimpses)
        if x_mask is not None:
            att = att.masked_fill(x_mask.squeeze(1).squeeze(1).unsqueeze(2), -1e9)
        att = F.softmax(att, dim=1)
        x_atted = torch.einsum("bng,bnd->bgd", att, x).reshape(x.shape[0], -1)  # glimpse-major concat (:104-110)
        return _linear(x_atted, self.linear_merge.weight, self.linear_merge.bias)


class AnswerModule(nn.Module):
    def __init__(self, num_answers, hidden_size=128, mcan_num_layers=4, mcan_num_heads=4, mcan_pdrop=0.1,
                 mcan_flat_mlp_size=512, mcan_flat_glimpses=1, mcan_flat_out_size=512):
        super().__init__()
        self.lang_feat_linear = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.GELU())
        self.object_feat_linear = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.GELU())
        self.object_cls = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.GELU(), nn.Dropout(0.1),
                                        nn.Linear(hidden_size, 1))
        self.answer_cls = nn.Sequential(nn.Linear(mcan_flat_out_size, hidden_size), nn.GELU(), nn.Dropout(0.1),
                                        nn.Linear(hidden_size, num_answers))
        self.attflat_visual = AttFlat(hidden_size, mcan_flat_mlp_size, mcan_flat_glimpses, mcan_flat_out_size, 0.1)
        self.attflat_lang = AttFlat(hidden_size, mcan_flat_mlp_size, mcan_flat_glimpses, mcan_flat_out_size, 0.1)

    def forward(self, data_dict):
        cross_feat = data_dict["cross_box_feature"]  # (B*L, K, hidden) from the match module
        fuse_feat = self.attflat_visual(cross_feat, None)
        a = self.answer_cls
        h = _linear(fuse_feat, a[0].weight, a[0].bias)
        if add_norm.act_dropout_supported(h) and not torch.is_autocast_enabled("cuda"):
            h = add_norm.act_dropout(h, "gelu", a[2].p, self.training)
        else:
            h = a[2](a[1](h))
        data_dict["answer_scores"] = _linear(h, a[3].weight, a[3].bias)  # (B*L, 128) x (num_answers, 128): library GEMM (counted)
        return data_dict

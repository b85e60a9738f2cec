"""Caption head of the joint model: `TransformerDecoderModel` as `models/jointnet/jointnet.py:104`
constructs it (`TransformerDecoderModel(30522)`), from models/caption_module/transformer_captioner.py — a 6-layer pre-norm
decoder (h = 8, d_model = 128, d_ff = 512) over [object indicator | caption tokens] with a vocabulary generator.  Same class
and attribute names, constructor arguments and state-dict keys (`model.decoder.layers.{i}.self_attn.linears.{0..3}`,
`…src_attn…` (constructed, unused with early_guide), `…feed_forward.w_{1,2}`, `…sublayer.{0,1,2}.norm.{a_2,b_2}`,
`model.decoder.norm`, `model.tgt_embed.0.lut`, `model.tgt_embed.1.pe`, `model.generator.proj`).

On the GPU the residual stream runs on csrc/add_norm.hip: ONE launch per sublayer boundary produces both
x + dropout(sublayer_out) and the NEXT sublayer's LayerNorm of it (the captioner's own LayerNorm, :117-129: unbiased std,
eps outside the root), the projections and w_1 on the MFMA linear kernels, ReLU + Dropout in one launch.  The attention
core (8 heads x 16 channels over <= 37 positions) and the 30 522-wide generator stay library calls (rocBLAS GEMMs).

Deviations from the file as shipped, all where it cannot run:
* the constructor there reads `lib/configs/config_caption.json` (absent from the reference tree; the value is never used)
  and downloads the bert-base-uncased tokenizer only for four constants — here `tokenizer` is optional and defaults to those
  constants (`BertUncasedIds`);
* `forward_train` with `caption_mlm=True` (the default) passes the (ids, mask) TUPLE returned by `mask()` to the embedding
  (:467-471) and raises; here the masked ids are fed (what `forward_mlm` :383-388 does);
* `mask()` mixes CPU probability tensors with CUDA ids (:599-617); here everything is drawn on the ids' device;
* `_prepare_feature` / `forward_*` call `.cuda()`; here tensors follow the inputs' device;
* `use_transformer_encoder=True` / `src_pos_type` (an encoder over the proposals; off in jointnet) are not built, and
  `early_guide=False` cannot run as shipped (`_prepare_feature` :366-381 always builds the mask for the sequence WITH the
  object indicator, one position longer than the late-guide decoder input): the constructor raises for all three.
"""
import copy
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import add_norm
from .mfma_linear import linear as _linear
from .nn_distance import nn_distance


class BertUncasedIds:
    """The four constants the captioner takes from BertTokenizer('bert-base-uncased')."""
    pad_token_id = 0
    cls_token_id = 101
    mask_token_id = 103
    vocab_size = 30522


def subsequent_mask(size):
    """(1, size, size) bool, True on and below the diagonal (:20-24)."""
    return torch.tril(torch.ones((1, size, size), dtype=torch.bool))


def clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def attention(query, key, value, mask=None, dropout=None):
    """softmax(q k^T / sqrt(d_k), masked_fill(mask == 0, -1e9)) v  (:32-42)."""
    d_k = query.size(-1)
    scores = torch.matmul(query, key.transpose(-2, -1)) / math.sqrt(d_k)
    if mask is not None:
        scores = scores.masked_fill(mask == 0, -1e9)
    p_attn = F.softmax(scores, dim=-1)
    if dropout is not None:
        p_attn = dropout(p_attn)
    return torch.matmul(p_attn, value), p_attn


class MultiHeadedAttention(nn.Module):
    """:45-78.  linears[0..2] project q / k / v, linears[3] the output."""

    def __init__(self, h, d_model, dropout=0.1, keep_value=False):
        super().__init__()
        assert d_model % h == 0
        self.d_k = d_model // h
        self.h = h
        self.linears = clones(nn.Linear(d_model, d_model), 4)
        self.attn = None
        self.dropout = nn.Dropout(p=dropout)
        self.keep_value = keep_value

    def forward(self, query, key, value, mask=None):
        if mask is not None:
            mask = mask.unsqueeze(1)  # same mask for all heads
        nb = query.size(0)
        query, key, value = [_linear(x, l.weight, l.bias).view(nb, -1, self.h, self.d_k).transpose(1, 2)
                             for l, x in zip(self.linears, (query, key, value))]
        x, self.attn = attention(query, key, value, mask=mask, dropout=self.dropout)
        if self.keep_value:
            self.value = value
        x = x.transpose(1, 2).contiguous().view(nb, -1, self.h * self.d_k)
        return _linear(x, self.linears[-1].weight, self.linears[-1].bias)


class PositionwiseFeedForward(nn.Module):
    """w_2(dropout(relu(w_1 x)))  (:81-91)."""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        z = _linear(x, self.w_1.weight, self.w_1.bias)
        if add_norm.act_dropout_supported(z) and not torch.is_autocast_enabled("cuda"):
            z = add_norm.act_dropout(z, "relu", self.dropout.p, self.training)
        else:
            z = self.dropout(F.relu(z))
        return _linear(z, self.w_2.weight, self.w_2.bias)


class Embeddings(nn.Module):
    """lut(x) * sqrt(d_model)  (:94-103)."""

    def __init__(self, d_model, vocab):
        super().__init__()
        self.lut = nn.Embedding(vocab, d_model)
        self.d_model = d_model

    def forward(self, x):
        return self.lut(x) * math.sqrt(self.d_model)


class Generator(nn.Module):
    """log_softmax(proj(x))  (:106-114)."""

    def __init__(self, d_model, vocab):
        super().__init__()
        self.proj = nn.Linear(d_model, vocab)

    def forward(self, x):
        return F.log_softmax(self.proj(x), dim=-1)


class LayerNorm(nn.Module):
    """a_2 * (x - mean) / (std + eps) + b_2 with torch's (unbiased) std  (:117-129)."""

    def __init__(self, features, eps=1e-6):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(features))
        self.b_2 = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x):
        if add_norm.sum_norm_supported(x):
            return add_norm.sum_norm(x, None, self.a_2, self.b_2, self.eps)[1]
        mean = x.mean(-1, keepdim=True)
        std = x.std(-1, keepdim=True)
        return self.a_2 * (x - mean) / (std + self.eps) + self.b_2


class SublayerConnection(nn.Module):
    """x + dropout(sublayer(norm(x)))  (:132-145).  Decoder.forward fuses the chain of these on the GPU."""

    def __init__(self, size, dropout):
        super().__init__()
        self.norm = LayerNorm(size)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, sublayer):
        return x + self.dropout(sublayer(self.norm(x)))


class PositionalEncoding(nn.Module):
    """Sinusoidal positions added to the embeddings, then dropout  (:148-167)."""

    def __init__(self, d_model, dropout, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1).float()
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0))

    def forward(self, x, src_pos=None):
        return self.dropout(x + self.pe[:, :x.size(1)])


class DecoderLayer(nn.Module):
    """self-attention, source attention (late guide only), feed forward — each inside a SublayerConnection  (:219-237)."""

    def __init__(self, size, self_attn, src_attn, feed_forward, dropout, early_guide=True):
        super().__init__()
        self.size = size
        self.self_attn = self_attn
        self.src_attn = src_attn
        self.feed_forward = feed_forward
        self.early_guide = early_guide
        self.sublayer = clones(SublayerConnection(size, dropout), 3)

    def forward(self, x, memory, src_mask, tgt_mask):
        m = memory
        x = self.sublayer[0](x, lambda x: self.self_attn(x, x, x, tgt_mask))
        if not self.early_guide:
            x = self.sublayer[1](x, lambda x: self.src_attn(x, m, m, src_mask))
        return self.sublayer[2](x, self.feed_forward)


class Decoder(nn.Module):
    """N layers and a final norm  (:202-216)."""

    def __init__(self, layer, N):
        super().__init__()
        self.layers = clones(layer, N)
        self.norm = LayerNorm(layer.size)

    def forward(self, x, memory, src_mask, tgt_mask, obj_indicator=None):
        if obj_indicator is not None:
            x = torch.cat((obj_indicator, x), dim=1)
        if not (add_norm.sum_norm_supported(x) and not torch.is_autocast_enabled("cuda")):
            for layer in self.layers:
                x = layer(x, memory, src_mask, tgt_mask)
            return self.norm(x)
        # fused residual stream: every boundary "x + dropout(sublayer_out)" and the NEXT norm in one launch
        x = x.contiguous()

        def boundary(s, y, dropout, norm):
            return add_norm.sum_norm(s, y, norm.a_2, norm.b_2, norm.eps, dropout.p if y is not None else 0.0,
                                     self.training)

        first = self.layers[0].sublayer[0]
        s, n = boundary(x, None, first.dropout, first.norm)
        for i, layer in enumerate(self.layers):
            sub = layer.sublayer
            a = layer.self_attn(n, n, n, tgt_mask)
            if not layer.early_guide:
                s, n = boundary(s, a, sub[0].dropout, sub[1].norm)
                b = layer.src_attn(n, memory, memory, src_mask)
                s, n = boundary(s, b, sub[1].dropout, sub[2].norm)
            else:
                s, n = boundary(s, a, sub[0].dropout, sub[2].norm)
            f = layer.feed_forward(n)
            nxt = self.layers[i + 1].sublayer[0].norm if i + 1 < len(self.layers) else self.norm
            s, n = boundary(s, f, sub[2].dropout, nxt)
        return n


class EncoderDecoder(nn.Module):
    """:240-283 with encoder = None (jointnet's configuration): src_embed is the identity and the proposals are the memory."""

    def __init__(self, encoder, decoder, src_embed, tgt_embed, generator, early_guide=True):
        super().__init__()
        if encoder is not None:
            raise NotImplementedError("the proposal encoder (use_transformer_encoder=True) is not built")
        self.encoder = encoder
        self.decoder = decoder
        self.src_embed = src_embed
        self.tgt_embed = tgt_embed
        self.generator = generator
        self.early_guide = early_guide

    def forward(self, src, tgt, src_mask, tgt_mask, obj_indicator=None, src_pos=None, obj_idx=None):
        return self.decode(self.src_embed(src, src_pos) if src_pos is not None else src, src_mask, tgt, tgt_mask,
                           obj_indicator=obj_indicator, obj_idx=obj_idx)

    def decode(self, memory, src_mask, tgt, tgt_mask, obj_indicator=None, obj_idx=None):
        if memory.shape[0] != tgt.shape[0]:  # inference: one caption per proposal
            assert memory.shape[0] * memory.shape[1] == tgt.shape[0]
            B, K, _ = memory.shape
            obj_indicator = obj_indicator + memory.view(B * K, -1).unsqueeze(1)
            if self.early_guide:
                memory = torch.repeat_interleave(memory, memory.shape[1], dim=0)
        if obj_idx is not None:
            obj_indicator = obj_indicator + torch.gather(memory, 1, obj_idx.repeat(1, memory.size(-1)).unsqueeze(1))
        if self.early_guide:
            return self.decoder(self.tgt_embed(tgt), memory, src_mask, tgt_mask, obj_indicator=obj_indicator)
        return self.decoder(self.tgt_embed(tgt), obj_indicator, None, tgt_mask, None)


def _identity(x, *_):
    return x


class TransformerDecoderModel(nn.Module):
    """:286-626.  forward(data_dict, is_eval) reads aggregated_vote_features (B,K,C), aggregated_vote_xyz (B,K,3),
    input_ids (B,L,T), ref_center_label_list (B,L,3), objectness_scores (B,K,2) and writes lang_cap, match_idx, pred_ious,
    good_bbox_masks (training) or lang_cap = greedy token ids (B,K,max_len+2) (evaluation)."""

    def make_model(self, tgt_vocab, N=6, h=8, d_model=128, d_ff=512, dropout=0.1, bn_momentum=0.1, src_pos_type=None,
                   use_transformer_encoder=False, early_guide=True):
        c = copy.deepcopy
        attn = MultiHeadedAttention(h, d_model)
        ff = PositionwiseFeedForward(d_model, d_ff, dropout)
        position = PositionalEncoding(d_model, dropout)
        model = EncoderDecoder(
            None,
            Decoder(DecoderLayer(d_model, c(attn), c(attn), c(ff), dropout, early_guide), N),
            _identity,
            nn.Sequential(Embeddings(d_model, tgt_vocab), c(position)),
            Generator(d_model, tgt_vocab), early_guide=early_guide)
        for p in model.parameters():  # Glorot / fan_avg
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        for m in model.modules():
            if isinstance(m, nn.BatchNorm1d):
                m.momentum = bn_momentum
        return model

    def __init__(self, vocab_size, N=6, h=8, d_model=128, d_ff=512, transformer_dropout=0.1, bn_momentum=0.1,
                 src_pos_type=None, use_transformer_encoder=False, early_guide=True, check_relation=False,
                 caption_mlm=True, tokenizer=None, max_des_len=36):
        super().__init__()
        if use_transformer_encoder or src_pos_type is not None:
            raise NotImplementedError("the proposal encoder / learned source positions are not built (off in jointnet)")
        if not early_guide:
            raise NotImplementedError("late guide is shape-inconsistent in the reference (mask one position longer than "
                                      "the decoder input, transformer_captioner.py:366-381): only early_guide=True")
        self.src_pos_type = src_pos_type
        self.use_transformer_encoder = use_transformer_encoder
        self.check_relation = check_relation
        self.early_guide = early_guide
        self.tokenizer = tokenizer if tokenizer is not None else BertUncasedIds()
        self.caption_mlm = caption_mlm
        self.vocab_size = vocab_size
        self.mask_ratio = 0.1
        self.max_des_len = max_des_len  # lib/configs/config_captioning.py:16 CONF.TRAIN.MAX_DES_LEN
        self.mlm_loss_fn = nn.CrossEntropyLoss(ignore_index=0, reduction="none")
        self.model = self.make_model(vocab_size, N=N, h=h, d_model=d_model, d_ff=d_ff, dropout=transformer_dropout,
                                     bn_momentum=bn_momentum, src_pos_type=src_pos_type,
                                     use_transformer_encoder=use_transformer_encoder, early_guide=early_guide)
        if check_relation:
            self.relation_proposal = nn.Sequential(nn.Linear(d_model, d_model), nn.ReLU(), nn.Linear(d_model, d_model),
                                                   nn.ReLU(), nn.Linear(d_model, 9))

    # ---- pieces shared by the three forward paths -------------------------------------------------------------
    def _prepare_feature(self, seq, captioning=True):
        """:366-381: drop the last token (and the first without early guide); mask = [1 | seq > 0] (& causal)."""
        seq = seq[:, :-1] if self.early_guide else seq[:, 1:-1]
        seq_mask = seq > 0
        seq_mask = torch.cat([torch.ones((seq_mask.shape[0], 1), dtype=torch.bool, device=seq.device), seq_mask], dim=1)
        seq_mask = seq_mask.unsqueeze(-2)
        if captioning:
            seq_mask = seq_mask & subsequent_mask(seq.size(-1) + 1).to(seq_mask.device)
        return seq, seq_mask

    def _reference_object(self, endpoints):
        """:393-416 / :447-463: the proposals repeated per sentence and the feature of the proposal nearest to each
        sentence's reference centre."""
        src = endpoints['aggregated_vote_features']
        input_ids = endpoints['input_ids']
        B, L, _ = input_ids.shape
        K = src.shape[1]
        input_ids = input_ids.view(B * L, -1)
        src = src[:, None, :, :].repeat(1, L, 1, 1).view(B * L, K, -1)
        vote_center = endpoints['aggregated_vote_xyz'][:, None, :, :].repeat(1, L, 1, 1).view(B * L, K, 3)
        ref_center = endpoints['ref_center_label_list'].view(B * L, -1)
        _, _, target_ious, idx = nn_distance(vote_center, ref_center.unsqueeze(1))
        endpoints['match_idx'] = idx.squeeze(1)
        ref_obj_feature = torch.gather(src, 1, idx.repeat(1, src.size(-1)).unsqueeze(1))  # (B*L, 1, C)
        return src, input_ids, target_ious, ref_obj_feature

    def mask(self, input_ids, vocab_size):
        """MLM corruption (:595-620): 10 % of the non-pad, non-[CLS] tokens; of those 80 % -> [MASK], 10 % -> a random
        word, 10 % unchanged.  Returns (ids, masked positions)."""
        ids = input_ids.clone()
        dev = ids.device
        masked = torch.bernoulli(torch.full(ids.shape, self.mask_ratio, device=dev)).bool()
        masked &= ids != self.tokenizer.pad_token_id
        masked &= ids != self.tokenizer.cls_token_id
        replaced = torch.bernoulli(torch.full(ids.shape, 0.8, device=dev)).bool() & masked
        ids = torch.where(replaced, torch.full_like(ids, self.tokenizer.mask_token_id), ids)
        rand = torch.bernoulli(torch.full(ids.shape, 0.5, device=dev)).bool() & masked & ~replaced
        ids = torch.where(rand, torch.randint(vocab_size, ids.shape, dtype=ids.dtype, device=dev), ids)
        return ids, masked

    def _decode_tokens(self, src, tokens, src_mask, seq_mask, ref_obj_feature):
        out = self.model(src=src, tgt=tokens, src_mask=src_mask.unsqueeze(1), tgt_mask=seq_mask,
                         obj_indicator=ref_obj_feature, src_pos=None, obj_idx=None)
        out = out[:, 1:, :] if self.early_guide else out  # drop the object-indicator position
        return self.model.generator(out)

    # ---- training ---------------------------------------------------------------------------------------------
    def forward_train(self, endpoints):
        """:431-492."""
        src, input_ids, target_ious, ref_obj_feature = self._reference_object(endpoints)
        seq, seq_mask = self._prepare_feature(input_ids)
        src_mask = endpoints["objectness_scores"].argmax(-1)
        tokens = self.mask(seq, self.tokenizer.vocab_size)[0] if self.caption_mlm else seq
        endpoints['lang_cap'] = self._decode_tokens(src, tokens, src_mask, seq_mask, ref_obj_feature)
        good = (target_ious > -1).squeeze(1)
        cnt = good.sum()
        endpoints["pred_ious"] = (target_ious.squeeze(1) * good).sum() / cnt.clamp(min=1)  # mean over good boxes, 0 if none
        endpoints["good_bbox_masks"] = good
        return endpoints

    def forward_mlm(self, endpoints):
        """:383-429: bidirectional (no causal mask) masked-token prediction and its loss."""
        src, input_ids, target_ious, ref_obj_feature = self._reference_object(endpoints)
        seq, seq_mask = self._prepare_feature(input_ids, captioning=False)
        src_mask = endpoints["objectness_scores"].argmax(-1)
        mask_seq, mask_index = self.mask(seq, self.tokenizer.vocab_size)
        pred = self._decode_tokens(src, mask_seq, src_mask, seq_mask, ref_obj_feature)
        endpoints['lang_mlm'] = pred
        num_words = pred.size(1)
        target = input_ids[:, 1:num_words + 1]
        loss = self.mlm_loss_fn(pred.reshape(-1, pred.shape[-1]), target.reshape(-1)) * mask_index.reshape(-1)
        good = (target_ious > -1).squeeze(1).unsqueeze(1).repeat(1, num_words).reshape(-1)
        endpoints["mlm_loss"] = torch.sum(loss * good) / (torch.sum(good) + 1e-6)
        return endpoints

    # ---- evaluation -------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_eval(self, endpoints):
        """:494-562: greedy decoding of one caption per proposal; every step re-runs the decoder on the prefix."""
        obj_features = endpoints["aggregated_vote_features"]
        B, K, _ = obj_features.shape
        src = torch.repeat_interleave(obj_features, K, dim=0)
        obj_features = obj_features.reshape(B * K, -1)
        ys = torch.full((B * K, 1), self.tokenizer.cls_token_id, dtype=torch.long, device=src.device)
        src_mask = endpoints["objectness_scores"].argmax(-1)
        for _ in range(self.max_des_len + 1):
            size = ys.size(1) + 1 if self.early_guide else ys.size(1)
            out = self.model(src=src, tgt=ys, src_mask=src_mask.unsqueeze(1),
                             tgt_mask=subsequent_mask(size).to(src.device), obj_indicator=obj_features.unsqueeze(1))
            prob = self.model.generator(out[:, -1, :])
            ys = torch.cat([ys, prob.argmax(dim=-1, keepdim=True)], dim=1)
        endpoints["lang_cap"] = ys.view(B, K, -1)
        return endpoints

    def forward(self, data_dict, is_eval=False):
        return self.forward_eval(data_dict) if is_eval else self.forward_train(data_dict)


def compute_cap_loss(data_dict, pad_token_id=0):
    """lib/loss_helper/loss_captioning.py:25-80 without its host synchronisations: token cross entropy (ignore_index 0)
    averaged over the tokens of good boxes, and the token accuracy over non-pad targets of good boxes (0 if none)."""
    pred = data_dict["lang_cap"]
    num_words, V = pred.size(1), pred.size(2)
    target = data_dict["input_ids"].view(pred.shape[0], -1)[:, 1:num_words + 1]
    loss = F.cross_entropy(pred.reshape(-1, V), target.reshape(-1), ignore_index=0, reduction="none")
    good = data_dict["good_bbox_masks"].unsqueeze(1).repeat(1, num_words).reshape(-1)
    cap_loss = torch.sum(loss * good) / (torch.sum(good) + 1e-6)
    valid = (target.reshape(-1) != pad_token_id) & good
    hit = (pred.reshape(-1, V).argmax(-1) == target.reshape(-1)) & valid
    cap_acc = hit.sum().float() / valid.sum().clamp(min=1).float()
    return cap_loss, cap_acc

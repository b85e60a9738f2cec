"""TEST INFRASTRUCTURE — CPU restatement of the reference's caption decoder, used only by tests/ as the checker.

Restates models/caption_module/transformer_captioner.py (the `TransformerDecoderModel(30522)` jointnet.py:104 builds:
N = 6, h = 8, d_model = 128, d_ff = 512, early_guide, no proposal encoder) as plain functions over a state dict, op by op
in the order of the reference, in the dtype of the given tensors (fp64 in the tests).  PARITY UNPINNED: the reference module
cannot be constructed here (its constructor reads lib/configs/config_caption.json, which the reference tree does not contain,
and downloads a tokenizer; its import needs `easydict`), so no fixture could be generated from it — this file is checked
against the source text only.
"""
import math

import torch
import torch.nn.functional as F


def layer_norm(x, a, b, eps=1e-6):
    """:117-129 — torch.std is the unbiased estimator; eps is added to the std, not the variance."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return a * (x - mean) / (std + eps) + b


def multi_head(sd, prefix, query, key, value, mask, h):
    """:45-78 + :32-42 (dropout off)."""
    nb, d_model = query.size(0), query.size(-1)
    d_k = d_model // h
    q, k, v = [F.linear(x, sd[f"{prefix}.linears.{i}.weight"], sd[f"{prefix}.linears.{i}.bias"])
               .view(nb, -1, h, d_k).transpose(1, 2) for i, x in enumerate((query, key, value))]
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(d_k)
    if mask is not None:
        scores = scores.masked_fill(mask.unsqueeze(1) == 0, -1e9)
    p = F.softmax(scores, dim=-1)
    x = torch.matmul(p, v).transpose(1, 2).contiguous().view(nb, -1, d_model)
    return F.linear(x, sd[f"{prefix}.linears.3.weight"], sd[f"{prefix}.linears.3.bias"])


def feed_forward(sd, prefix, x):
    """:81-91."""
    return F.linear(F.relu(F.linear(x, sd[f"{prefix}.w_1.weight"], sd[f"{prefix}.w_1.bias"])),
                    sd[f"{prefix}.w_2.weight"], sd[f"{prefix}.w_2.bias"])


def positional_encoding(T, d_model, dtype):
    """:148-163."""
    pe = torch.zeros(T, d_model, dtype=torch.float32)
    position = torch.arange(0, T).unsqueeze(1).float()
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.to(dtype)


def decode(sd, tokens, obj_indicator, tgt_mask, N=6, h=8, early_guide=True):
    """EncoderDecoder.decode :262-283 -> Decoder.forward :209-216 -> DecoderLayer.forward :231-237 (eval mode).
    tokens (n, T) int64, obj_indicator (n, 1, C), tgt_mask (n, 1|T+1, T+1) bool -> (n, T+1, C)."""
    lut = sd["model.tgt_embed.0.lut.weight"]
    d_model = lut.shape[1]
    x = lut[tokens] * math.sqrt(d_model)                                         # :102
    x = x + positional_encoding(x.size(1), d_model, x.dtype).unsqueeze(0)         # :165
    if early_guide:
        x = torch.cat((obj_indicator, x), dim=1)                                  # :210-211
        memory = None
    else:
        memory = obj_indicator                                                    # :283
    for i in range(N):
        p = f"model.decoder.layers.{i}"
        n = layer_norm(x, sd[f"{p}.sublayer.0.norm.a_2"], sd[f"{p}.sublayer.0.norm.b_2"])
        x = x + multi_head(sd, f"{p}.self_attn", n, n, n, tgt_mask, h)            # :233
        if not early_guide:
            n = layer_norm(x, sd[f"{p}.sublayer.1.norm.a_2"], sd[f"{p}.sublayer.1.norm.b_2"])
            x = x + multi_head(sd, f"{p}.src_attn", n, memory, memory, None, h)   # :235
        n = layer_norm(x, sd[f"{p}.sublayer.2.norm.a_2"], sd[f"{p}.sublayer.2.norm.b_2"])
        x = x + feed_forward(sd, f"{p}.feed_forward", n)                          # :237
    return layer_norm(x, sd["model.decoder.norm.a_2"], sd["model.decoder.norm.b_2"])


def forward_train(sd, endpoints, N=6, h=8, early_guide=True):
    """forward_train :431-492 with caption_mlm off (eval-mode arithmetic): returns lang_cap (B*L, T-1, V), match_idx."""
    src = endpoints['aggregated_vote_features']
    input_ids = endpoints['input_ids']
    B, L, _ = input_ids.shape
    K = src.shape[1]
    input_ids = input_ids.view(B * L, -1)
    src = src[:, None].repeat(1, L, 1, 1).view(B * L, K, -1)
    vote_center = endpoints['aggregated_vote_xyz'][:, None].repeat(1, L, 1, 1).view(B * L, K, 3)
    ref_center = endpoints['ref_center_label_list'].view(B * L, -1)
    d = ((vote_center - ref_center.unsqueeze(1)) ** 2).sum(-1)                    # nn_distance :447-456 (L2 squared)
    idx = d.argmin(dim=1, keepdim=True)
    ref_obj = torch.gather(src, 1, idx.repeat(1, src.size(-1)).unsqueeze(1))
    seq = input_ids[:, :-1] if early_guide else input_ids[:, 1:-1]                # :366-381
    seq_mask = torch.cat([torch.ones(seq.shape[0], 1, dtype=torch.bool), seq > 0], dim=1).unsqueeze(-2)
    T1 = seq.size(-1) + 1
    seq_mask = seq_mask & torch.tril(torch.ones(1, T1, T1, dtype=torch.bool))
    out = decode(sd, seq, ref_obj, seq_mask, N, h, early_guide)
    out = out[:, 1:, :] if early_guide else out
    logits = F.linear(out, sd["model.generator.proj.weight"], sd["model.generator.proj.bias"])
    return F.log_softmax(logits, dim=-1), idx.squeeze(1)


def cap_loss(lang_cap, input_ids, good):
    """lib/loss_helper/loss_captioning.py:25-48."""
    num_words, V = lang_cap.size(1), lang_cap.size(2)
    target = input_ids.view(lang_cap.shape[0], -1)[:, 1:num_words + 1]
    loss = F.cross_entropy(lang_cap.reshape(-1, V), target.reshape(-1), ignore_index=0, reduction="none")
    g = good.unsqueeze(1).repeat(1, num_words).reshape(-1)
    return torch.sum(loss * g) / (torch.sum(g) + 1e-6)

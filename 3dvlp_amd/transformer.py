"""Attention blocks of the grounding path with the reference's interface and state_dict layout.

Mirrors models/transformer/attention.py (ScaledDotProductAttention :6-78, MultiHeadAttention :81-131)
and models/transformer/mmattention.py (PositionwiseFeedForward :36-50, CrossAttentionDecoderLayer
:53-86).  The projections / FFN are plain library GEMMs; the softmax(QK^T/sqrt(dk) [+bias]) V core
runs in the fused HIP kernel (``impl='hip'``, the default: no (B,h,nq,nk) matrix is materialised).
``impl='torch'`` is the unfused formulation, kept ONLY as an explicit opt-in for host-logic unit
tests and A/B numerics — it is never selected automatically.
"""
import os

import numpy as np
import torch
from torch import nn

from . import add_norm, fused_attention, mfma_linear, row_chain
from .ddp import merge_adjacent
from .mfma_linear import linear as _linear

DEFAULT_IMPL = "hip"
# The match decoder's attention cores on bf16 ROWS (SURVEY.md §8(d)'s bytes: q, k, v, out touched once as bf16): the row chains
# / the query projection store what the cores read as bf16 and the cores store what the chains read as bf16 — the values the
# bf16-MFMA kernels round their operands to anyway.  "0": fp32 rows between the launches (round 3's form).
ATTN_BF16_ROWS = os.environ.get("VLP3D_ATTN_BF16_ROWS", "1") != "0"
# The decoder chains keep of their FFN stage only its bf16 output rows (row_chain._Chain compact_acts): no pre-activation, no fp32 h
CHAIN_COMPACT_ACTS = os.environ.get("VLP3D_CHAIN_COMPACT", "1") != "0"


class ScaledDotProductAttention(nn.Module):
    """out = fc_o(softmax(fc_q(q) fc_k(k)^T / sqrt(d_k) [+|* weights] [masked]) fc_v(v))."""

    def __init__(self, d_model, d_k, d_v, h, impl=None):
        super().__init__()
        self.fc_q = nn.Linear(d_model, h * d_k)
        self.fc_k = nn.Linear(d_model, h * d_k)
        self.fc_v = nn.Linear(d_model, h * d_v)
        self.fc_o = nn.Linear(h * d_v, d_model)
        self.d_model, self.d_k, self.d_v, self.h = d_model, d_k, d_v, h
        self.impl = impl
        self.bf16_mma = False  # True: bf16 MFMA operands in the fused core (timing configuration of the step driver)
        self.merge_qkv = True  # one projection launch for inputs shared by q/k/v (self) or k/v (cross attention)
        for fc in (self.fc_q, self.fc_k, self.fc_v, self.fc_o):
            nn.init.xavier_uniform_(fc.weight)
            nn.init.constant_(fc.bias, 0)

    def forward(self, queries, keys, values, attention_mask=None, attention_weights=None, way="add",
                need_att=True, with_residual=False, project=True):
        """queries (b,nq,d_model), keys/values (b,nk,d_model); attention_mask broadcastable to
        (b,h,nq,nk) with 0 = masked (filled with -10000); attention_weights (b,h,nq,nk).
        Returns (out (b,nq,d_model), att (b,h,nq,nk) or None when the fused kernel ran).
        with_residual: returns (out, att, q_res) — q_res is `queries` routed through the query projection's autograd
        node (mfma_linear.linear(with_residual=True)), for the residual connection of the caller.
        project=False: `out` is the heads' concatenated output BEFORE fc_o (the caller runs fc_o itself, e.g. as the first
        stage of a row chain together with the add & norm that follows)."""
        if with_residual:
            res = self._forward(queries, keys, values, attention_mask, attention_weights, way, need_att, True, project)
            return res if len(res) == 3 else (res[0], res[1], queries)
        return self._forward(queries, keys, values, attention_mask, attention_weights, way, need_att, False, project)

    def _forward(self, queries, keys, values, attention_mask, attention_weights, way, need_att, wr, project=True):
        b_s, nq = queries.shape[:2]
        nk = keys.shape[1]
        if way not in ("add", "mul"):
            raise NotImplementedError(way)
        impl = self.impl or DEFAULT_IMPL
        fused = impl == "hip" and not need_att and fused_attention.supported(self.d_k, self.d_v, attention_mask, nk, b_s)
        if fused and self.merge_qkv and keys is values and queries.is_cuda and queries.dtype == torch.float32:
            # projections that read the same input run as ONE linear layer over concatenated weights (the parameters
            # keep the reference's names): q|k|v for self-attention, k|v for cross-attention; the attention kernels
            # take the column blocks as row-strided views and return one merged gradient
            aw = None if attention_weights is None else attention_weights.float()
            q_res = None
            if queries is keys:
                qkv = _linear(queries, merge_adjacent([self.fc_q.weight, self.fc_k.weight, self.fc_v.weight]),
                              merge_adjacent([self.fc_q.bias, self.fc_k.bias, self.fc_v.bias]), with_residual=wr)
                if wr:
                    qkv, q_res = qkv
                out = fused_attention.sdpa_merged(qkv, None, self.h, aw, way, attention_mask, bf16_mma=self.bf16_mma)
            else:
                q = _linear(queries, self.fc_q.weight, self.fc_q.bias, with_residual=wr)
                if wr:
                    q, q_res = q
                kv = _linear(keys, merge_adjacent([self.fc_k.weight, self.fc_v.weight]),
                             merge_adjacent([self.fc_k.bias, self.fc_v.bias]))
                out = fused_attention.sdpa_merged(q, kv, self.h, aw, way, attention_mask, bf16_mma=self.bf16_mma)
            o = _linear(out, self.fc_o.weight, self.fc_o.bias) if project else out
            return (o, None, q_res) if wr else (o, None)
        q = _linear(queries, self.fc_q.weight, self.fc_q.bias)
        k = _linear(keys, self.fc_k.weight, self.fc_k.bias)
        v = _linear(values, self.fc_v.weight, self.fc_v.bias)
        if fused:
            out = fused_attention.sdpa(q.float(), k.float(), v.float(), self.h,
                                       None if attention_weights is None else attention_weights.float(), way,
                                       attention_mask, bf16_mma=self.bf16_mma)
            return (_linear(out, self.fc_o.weight, self.fc_o.bias) if project else out), None
        if impl == "hip" and not q.is_cuda:
            raise RuntimeError("CPU not supported (impl='hip'); pass impl='torch' explicitly for host-side tests")
        q = q.view(b_s, nq, self.h, self.d_k).permute(0, 2, 1, 3)
        k = k.view(b_s, nk, self.h, self.d_k).permute(0, 2, 3, 1)
        v = v.view(b_s, nk, self.h, self.d_v).permute(0, 2, 1, 3)
        att = torch.matmul(q, k) / np.sqrt(self.d_k)
        if attention_weights is not None:
            att = att * attention_weights if way == "mul" else att + attention_weights
        if attention_mask is not None:
            att = att.masked_fill(attention_mask == 0, -10000)
        att = torch.softmax(att, -1)
        out = torch.matmul(att, v).permute(0, 2, 1, 3).contiguous().view(b_s, nq, self.h * self.d_v)
        return (_linear(out, self.fc_o.weight, self.fc_o.bias) if project else out), att


class MultiHeadAttention(nn.Module):
    """SDPA -> dropout -> LayerNorm(queries + out)  (post-LN; attention.py:108-131).
    `identity_map_reordering` / stateful decoding are not used on the grounding path."""

    def __init__(self, d_model, d_k, d_v, h, dropout=.1, identity_map_reordering=False, impl=None):
        super().__init__()
        self.identity_map_reordering = identity_map_reordering
        self.attention = ScaledDotProductAttention(d_model=d_model, d_k=d_k, d_v=d_v, h=h, impl=impl)
        self.dropout = nn.Dropout(p=dropout)
        self.layer_norm = nn.LayerNorm(d_model)
        self.fused_norm = False  # True: dropout + residual + LayerNorm in one HIP kernel (add_norm.py); the owner of
        # the training loop must then call add_norm.advance(device) once per step (GroundingStep does)

    def forward(self, queries, keys, values, attention_mask=None, attention_weights=None, way="add",
                output_attn=False):
        if self.identity_map_reordering:
            q_norm, k_norm, v_norm = (self.layer_norm(t) for t in (queries, keys, values))
            out, att = self.attention(q_norm, k_norm, v_norm, attention_mask, attention_weights, way,
                                      need_att=output_attn)
            out = queries + self.dropout(torch.relu(out))
        else:
            fo = self.attention.fc_o
            chain = [row_chain.linear_add_norm(fo.weight, fo.bias, self.layer_norm, queries, self.dropout.p)]
            if (self.fused_norm and not output_attn and queries.dim() == 3
                    and row_chain.supported(queries, chain, rows=queries.shape[0] * queries.shape[1])):
                # fc_o -> dropout -> add -> LayerNorm as ONE launch each way (row_chain.py; bf16 configuration)
                a, _, q_res = self.attention(queries, keys, values, attention_mask, attention_weights, way, need_att=False,
                                             with_residual=True, project=False)
                chain[0]["res"] = q_res
                return row_chain.run(a, chain, self.training)[0].view(queries.shape)
            out, att, q_res = self.attention(queries, keys, values, attention_mask, attention_weights, way,
                                             need_att=output_attn, with_residual=True)
            if self.fused_norm and add_norm.supported(queries, out, self.layer_norm):
                out = add_norm.add_norm(q_res, out, self.layer_norm, self.dropout.p, self.training)
            else:
                out = self.layer_norm(q_res + self.dropout(out))
        return (out, att) if output_attn else out


class PositionwiseFeedForward(nn.Module):
    def __init__(self, d_model, hidden, drop_prob=0.1):
        super().__init__()
        self.linear1 = nn.Linear(d_model, hidden)
        self.linear2 = nn.Linear(hidden, d_model)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(p=drop_prob)

    def forward(self, x, with_residual=False):
        """with_residual: returns (y, x_res) — x routed through linear1's autograd node for the caller's residual add."""
        z = _linear(x, self.linear1.weight, self.linear1.bias, with_residual=with_residual)
        if with_residual:
            z, x_res = z
        if add_norm.act_dropout_supported(z) and not torch.is_autocast_enabled("cuda"):
            h = add_norm.act_dropout(z, "relu", self.dropout.p, self.training)  # relu + dropout: one launch each way
        else:
            h = self.dropout(self.relu(z))
        y = _linear(h, self.linear2.weight, self.linear2.bias)
        return (y, x_res) if with_residual else y


class CrossAttentionDecoderLayer(nn.Module):
    """self-attention -> proposal<->token cross-attention -> FFN, each followed by add & norm."""

    def __init__(self, ffn_hidden=256, hidden_size=128, head=4, drop_prob=.1, impl=None):
        super().__init__()
        dk = hidden_size // head
        self.self_attention = MultiHeadAttention(d_model=hidden_size, d_k=dk, d_v=dk, h=head, impl=impl)
        self.enc_dec_attention = MultiHeadAttention(d_model=hidden_size, d_k=dk, d_v=dk, h=head, impl=impl)
        self.ffn = PositionwiseFeedForward(d_model=hidden_size, hidden=ffn_hidden, drop_prob=drop_prob)
        self.norm = nn.LayerNorm(hidden_size)
        self.dropout = nn.Dropout(p=drop_prob)
        self.head = head
        self.fused_norm = False  # see MultiHeadAttention.fused_norm

    def forward(self, query, key, value, src_mask=None, src_trg_mask=None):
        x = self.self_attention(query, query, query, attention_mask=src_mask)
        return self._after_self_attention(x, key, value, src_trg_mask)

    def _after_self_attention(self, x, key, value, src_trg_mask):
        x = self.enc_dec_attention(x, key, value, attention_mask=src_trg_mask)
        f, x_res = self.ffn(x, with_residual=True)
        if self.fused_norm and add_norm.supported(x, f, self.norm):
            return add_norm.add_norm(x_res, f, self.norm, self.dropout.p, self.training)
        return self.norm(self.dropout(f) + x_res)

    def forward_tiled(self, query, rep, key, value, src_trg_mask=None):
        """== forward(query tiled `rep` times along the batch, key, value) for a query (B, K, C) that `rep` consecutive
        key / value sequences share (match_module.py:127-137 tiles the proposals over a scene's sentences and feeds the copies
        to the first decoder layer).  The attention block of the self-attention sublayer (attention.py:41-78: projections,
        softmax, fc_o — no dropout inside) is identical for all copies, so it runs ONCE on the B sequences; what differs per
        copy is the dropout mask of the add & norm that follows (attention.py:128-130), which a replicating add & norm kernel
        draws per copy.  8x fewer rows through q|k|v, the attention core and fc_o (forward and backward) at L = 8, and the
        16.8 MB tiled copy of the proposal features is never made."""
        mha = self.self_attention
        B, K, C = query.shape
        if (rep == 1 or not self.fused_norm or not mha.fused_norm or mha.identity_map_reordering or not query.is_cuda
                or query.dtype != torch.float32 or C not in (64, 128, 256) or torch.is_autocast_enabled("cuda")):
            tiled = query[:, None].expand(B, rep, K, C).reshape(B * rep, K, C)
            return self.forward(tiled, key, value, src_trg_mask=src_trg_mask)
        out, _, q_res = mha.attention(query, query, query, need_att=False, with_residual=True)
        x = add_norm.add_norm_rep(q_res, out, mha.layer_norm, rep, mha.dropout.p, mha.training)
        return self._after_self_attention(x, key, value, src_trg_mask)


def _attn_weights(att):
    return (merge_adjacent([att.fc_q.weight, att.fc_k.weight, att.fc_v.weight]),
            merge_adjacent([att.fc_q.bias, att.fc_k.bias, att.fc_v.bias]))


def decoder_stack_chained(layers, query, rep, key, tail=()):
    """The decoder stack of match_module.py:127-137 — `layers[0].forward_tiled(query, rep, key, key)` followed by
    `layer(x, key, key)` for the other layers — with every row-local run between two attention cores as ONE launch
    (row_chain.py): [fc_o -> add & norm -> next fc_q] after a self-attention core, [fc_o -> add & norm -> FFN -> add & norm ->
    the next layer's fc_q|k|v, or the `tail` stages after the last layer] after a cross-attention core.
    tail: row_chain stages applied to the stack's output rows (match_module.py:40-47's Linear/GELU/Dropout pairs).
    Returns (x (B*rep, K, C), tail output (B*rep*K, N) or None), or None when a shape / mode is outside what the chain kernel
    covers (the caller then runs the layer modules)."""
    l0 = layers[0]
    mha = l0.self_attention
    B, K, C = query.shape
    R = B * rep * K
    if (not query.is_cuda or query.dtype != torch.float32 or C != 128 or not l0.fused_norm or not mha.fused_norm
            or mha.identity_map_reordering or not mfma_linear.BF16_MMA or R % 64 or torch.is_autocast_enabled("cuda")):
        return None
    probe = query  # device / dtype / width of the chains' inputs; the row count goes with `rows=`
    plan = []
    for i, layer in enumerate(layers):
        ca, ffn = layer.enc_dec_attention, layer.ffn
        p = layer.dropout.p
        st = [row_chain.linear_add_norm(ca.attention.fc_o.weight, ca.attention.fc_o.bias, ca.layer_norm, probe, ca.dropout.p),
              row_chain.linear(ffn.linear1.weight, ffn.linear1.bias, "relu", ffn.dropout.p),
              row_chain.linear_add_norm(ffn.linear2.weight, ffn.linear2.bias, layer.norm, ("tile", 1), p)]
        if i + 1 < len(layers):
            w, b = _attn_weights(layers[i + 1].self_attention.attention)
            st.append(row_chain.linear(w, b))
        else:
            st.extend(tail)
        plan.append(st)
        if not row_chain.supported(probe, st, rows=R):
            return None
        if i > 0:
            sa = layer.self_attention
            head = [row_chain.linear_add_norm(sa.attention.fc_o.weight, sa.attention.fc_o.bias, sa.layer_norm, probe, sa.dropout.p),
                    row_chain.linear(ca.attention.fc_q.weight, ca.attention.fc_q.bias)]
            if not row_chain.supported(probe, head, rows=R):
                return None
    h = mha.attention.h
    bf = mha.attention.bf16_mma
    # layer 0: the copy-independent self-attention block on the B sequences, replicated by its add & norm (forward_tiled)
    out, _, q_res = mha.attention(query, query, query, need_att=False, with_residual=True)
    x = add_norm.add_norm_rep(q_res, out, mha.layer_norm, rep, mha.dropout.p, mha.training).reshape(R, C)
    ca = l0.enc_dec_attention.attention
    # bf16 rows between the chains and the cores (ATTN_BF16_ROWS): q / q|k|v / the cores' outputs exist as bf16 rows only; the
    # fp32 tensors of the same names are shells that carry the autograd edges (fused_attention._SDPARows)
    rows = (ATTN_BF16_ROWS and bf and h * 32 == C and fused_attention.rows_supported(K, max(K, key.shape[1]), B * rep * h)
            and mfma_linear.rows16_supported(x, ca.fc_q.weight))
    q_rows = None
    if rows:
        q, x, q_rows = mfma_linear.linear_rows16(x, ca.fc_q.weight, ca.fc_q.bias)
    else:
        q, x = _linear(x, ca.fc_q.weight, ca.fc_q.bias, with_residual=True)
    tail_out = None
    for i, layer in enumerate(layers):
        ca = layer.enc_dec_attention.attention
        more = i + 1 < len(layers)
        wkv, bkv = merge_adjacent([ca.fc_k.weight, ca.fc_v.weight]), merge_adjacent([ca.fc_k.bias, ca.fc_v.bias])
        kv_rows = None
        if rows and mfma_linear.rows16_supported(key, wkv):  # the tokens' k | v as bf16 rows too (row count a multiple of 32)
            kv, kv_rows = mfma_linear.linear_rows16(key, wkv, bkv, with_residual=False)
        else:
            kv = _linear(key, wkv, bkv)
        st = plan[i]
        st[0]["res"] = x
        if rows:
            a, a_rows = fused_attention.sdpa_rows(q.view(B * rep, K, C), q_rows.view(B * rep, K, C), kv, h, b_rows=kv_rows)
            t = row_chain.run(a.reshape(R, C), st, layer.training, x_rows=a_rows, last_rows=more, compact_acts=CHAIN_COMPACT_ACTS)
        else:
            a = fused_attention.sdpa_merged(q.view(B * rep, K, C), kv, h, None, "add", None, bf16_mma=bf)
            t = row_chain.run(a.reshape(R, C), st, layer.training, compact_acts=CHAIN_COMPACT_ACTS)
        x = t[2]
        if more:
            nl = layers[i + 1]
            sa, nca = nl.self_attention, nl.enc_dec_attention.attention
            head = [row_chain.linear_add_norm(sa.attention.fc_o.weight, sa.attention.fc_o.bias, sa.layer_norm, x, sa.dropout.p),
                    row_chain.linear(nca.fc_q.weight, nca.fc_q.bias)]
            if rows:
                a, a_rows = fused_attention.sdpa_rows(t[3].view(B * rep, K, 3 * C), t[4].view(B * rep, K, 3 * C), None, h)
                x, q, q_rows = row_chain.run(a.reshape(R, C), head, nl.training, x_rows=a_rows, last_rows=True)
            else:
                a = fused_attention.sdpa_merged(t[3].view(B * rep, K, 3 * C), None, h, None, "add", None, bf16_mma=bf)
                x, q = row_chain.run(a.reshape(R, C), head, nl.training)
        elif tail:
            tail_out = t[-1]
    return x.view(B * rep, K, C), tail_out


"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU baseline for bench.py's `cpu_baseline` leg.

Times the oracle's FORWARD pass of the grounding path on ONE synthetic scene of the bench workload
(40 000 points, 256 proposals, 8 sentences): geometry (FPS / ball query / group / three_nn /
interpolate) in the OpenMP C restatement, dense layers (SharedMLP, attention) in numpy.  The oracle
has no backward, so the figure is forward-only and says so; it is a reported baseline, not a target.
"""
import os
import time

import numpy as np

from . import oracle as orc


def _layers(rng, dims):
    return [dict(w=rng.normal(0, 0.1, (dims[i + 1], dims[i])).astype(np.float32), gamma=np.ones(dims[i + 1]),
                 beta=np.zeros(dims[i + 1]), mean=np.zeros(dims[i + 1]), var=np.ones(dims[i + 1]))
            for i in range(len(dims) - 1)]


def _mha_weights(rng, prefix=""):
    W = {}
    for n in ("q", "k", "v", "o"):
        W[f"{prefix}attention.fc_{n}.weight"] = rng.normal(0, 0.08, (128, 128))
        W[f"{prefix}attention.fc_{n}.bias"] = np.zeros(128)
    W[f"{prefix}layer_norm.weight"], W[f"{prefix}layer_norm.bias"] = np.ones(128), np.zeros(128)
    return W


def scene_forward(xyz, feats, lang_num=8, seed=0):
    """xyz (1,N,3), feats (1,C,N). Returns (seconds, breakdown dict)."""
    rng = np.random.default_rng(seed)
    t = {}
    t0 = time.perf_counter()
    cfg = [(2048, 0.2, 64, [feats.shape[1] + 3, 64, 64, 128]), (1024, 0.4, 32, [131, 128, 128, 256]),
           (512, 0.8, 16, [259, 128, 128, 256]), (256, 1.2, 16, [259, 128, 128, 256])]
    cur_xyz, cur_f = xyz, feats
    levels = []
    for i, (m, r, ns, dims) in enumerate(cfg):
        s = time.perf_counter()
        cur_xyz, cur_f, _ = orc.sa_module_votes(cur_xyz, cur_f, _layers(rng, dims), m, r, ns, training=True)
        levels.append((cur_xyz, cur_f))
        t[f"sa{i + 1}"] = time.perf_counter() - s
    s = time.perf_counter()
    f = orc.fp_module(levels[2][0], levels[3][0], levels[2][1], levels[3][1], _layers(rng, [512, 256, 256]), True)
    f = orc.fp_module(levels[1][0], levels[2][0], levels[1][1], f, _layers(rng, [512, 256, 256]), True)
    t["fp"] = time.perf_counter() - s
    s = time.perf_counter()
    vote_xyz = levels[1][0] + rng.normal(0, 0.05, levels[1][0].shape).astype(np.float32)
    agg_xyz, agg_f, _ = orc.sa_module_votes(vote_xyz, f, _layers(rng, [259, 128, 128, 128]), 256, 0.3, 16, True)
    t["vote_agg"] = time.perf_counter() - s
    s = time.perf_counter()
    x = agg_f.transpose(0, 2, 1).astype(np.float64)
    for _ in range(2):  # relation self-attention
        x, _ = orc.multi_head_attention(_mha_weights(rng), x, x, x, 4)
    xq = np.repeat(x, lang_num, axis=0)
    lang = rng.normal(size=(lang_num, 49, 128))
    for _ in range(2):  # match decoder layers
        W = {}
        W.update(_mha_weights(rng, "self_attention."))
        W.update(_mha_weights(rng, "enc_dec_attention."))
        W.update({"ffn.linear1.weight": rng.normal(0, 0.08, (256, 128)), "ffn.linear1.bias": np.zeros(256),
                  "ffn.linear2.weight": rng.normal(0, 0.08, (128, 256)), "ffn.linear2.bias": np.zeros(128),
                  "norm.weight": np.ones(128), "norm.bias": np.zeros(128)})
        xq = orc.cross_attention_decoder_layer(W, xq, lang, lang)
    t["attention"] = time.perf_counter() - s
    s = time.perf_counter()
    orc.nn_distance(agg_xyz, agg_xyz)
    t["nn_distance"] = time.perf_counter() - s
    return time.perf_counter() - t0, t


def threads_used():
    """Threads the C/OpenMP part actually runs on (numpy's BLAS may use its own pool for the dense part)."""
    return int(orc.lib().orc_num_threads())

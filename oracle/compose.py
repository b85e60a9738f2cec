"""ORACLE — TEST INFRASTRUCTURE ONLY.  The detection trunk composed from the oracle's pieces.

Restates, on numpy arrays and with weights taken from a ``state_dict`` (numpy values, reference key names):

  Pointnet2Backbone.forward   models/base_module/backbone_module.py:76-135
  VotingModule.forward        models/base_module/voting_module.py:33-60
  L2 normalisation of votes   models/jointnet/jointnet.py:148-149
  vote aggregation            models/proposal_module/proposal_module_fcos.py:36-43,76

The nine native ops inside come from pointnet2_oracle.c ("parity unpinned", see its header); the dense layers are
the fixture-pinned numpy restatements of oracle.py (SharedMLP, VotingModule).  Train-mode BatchNorm uses batch
statistics over the whole batch, so callers must hand over the same scenes the model saw.
"""
import numpy as np

from . import oracle as orc

SA_CFG = (("sa1", 2048, 0.2, 64), ("sa2", 1024, 0.4, 32), ("sa3", 512, 0.8, 16), ("sa4", 256, 1.2, 16))


def mlp_layers(W, prefix):
    """SharedMLP weights under `prefix` (lib/pointnet2/pytorch_utils.py:11-36 key layout) -> oracle.shared_mlp layers."""
    out, i = [], 0
    while f"{prefix}layer{i}.conv.weight" in W:
        p = f"{prefix}layer{i}."
        w = np.asarray(W[p + "conv.weight"])
        out.append(dict(w=w.reshape(w.shape[0], w.shape[1]), gamma=W[p + "bn.bn.weight"], beta=W[p + "bn.bn.bias"],
                        mean=W[p + "bn.bn.running_mean"], var=W[p + "bn.bn.running_var"]))
        i += 1
    return out


def _conv1d(x, w, b):
    w = np.asarray(w, np.float64)
    return np.einsum("oc,bcn->bon", w.reshape(w.shape[0], w.shape[1]), x) + np.asarray(b, np.float64)[None, :, None]


def _bn1d(x, W, prefix, training, eps=1e-5):
    if training:
        mean, var = x.mean(axis=(0, 2)), x.var(axis=(0, 2))
    else:
        mean, var = np.asarray(W[prefix + "running_mean"], np.float64), np.asarray(W[prefix + "running_var"], np.float64)
    y = (x - mean[None, :, None]) / np.sqrt(var[None, :, None] + eps)
    return y * np.asarray(W[prefix + "weight"], np.float64)[None, :, None] + \
        np.asarray(W[prefix + "bias"], np.float64)[None, :, None]


def voting_module(W, prefix, seed_xyz, seed_features, training, vote_factor=1):
    """voting_module.py:33-60.  seed_xyz (B,S,3), seed_features (B,C,S) -> vote_xyz (B,S*vf,3), vote_features (B,C,S*vf)."""
    x = np.asarray(seed_features, np.float64)
    B, C, S = x.shape
    net = np.maximum(_bn1d(_conv1d(x, W[prefix + "conv1.weight"], W[prefix + "conv1.bias"]), W, prefix + "bn1.", training), 0)
    net = np.maximum(_bn1d(_conv1d(net, W[prefix + "conv2.weight"], W[prefix + "conv2.bias"]), W, prefix + "bn2.", training), 0)
    net = _conv1d(net, W[prefix + "conv3.weight"], W[prefix + "conv3.bias"])
    net = net.transpose(0, 2, 1).reshape(B, S, vote_factor, 3 + C)
    vote_xyz = (np.asarray(seed_xyz, np.float64)[:, :, None, :] + net[..., 0:3]).reshape(B, S * vote_factor, 3)
    vote_features = x.transpose(0, 2, 1)[:, :, None, :] + net[..., 3:]
    vote_features = vote_features.reshape(B, S * vote_factor, C).transpose(0, 2, 1)
    return vote_xyz.astype(np.float32), vote_features.astype(np.float32)


def detection_trunk(W, point_clouds, training=True, backbone="backbone_net.", vgen="vgen.",
                    vote_agg="proposal.vote_aggregation.", num_proposal=256):
    """point_clouds (B,N,3+C) fp32 -> dict with the reference's data_dict keys up to aggregated_vote_features."""
    pc = np.asarray(point_clouds, np.float32)
    xyz = np.ascontiguousarray(pc[..., :3])
    features = np.ascontiguousarray(pc[..., 3:].transpose(0, 2, 1))
    d = {}
    for name, npoint, radius, nsample in SA_CFG:
        xyz, features, inds = orc.sa_module_votes(xyz, features, mlp_layers(W, f"{backbone}{name}.mlp_module."), npoint,
                                                  radius, nsample, training, normalize_xyz=True)
        d[name + "_inds"], d[name + "_xyz"], d[name + "_features"] = inds, xyz, features
    f = orc.fp_module(d["sa3_xyz"], d["sa4_xyz"], d["sa3_features"], d["sa4_features"],
                      mlp_layers(W, backbone + "fp1.mlp."), training)
    f = orc.fp_module(d["sa2_xyz"], d["sa3_xyz"], d["sa2_features"], f, mlp_layers(W, backbone + "fp2.mlp."), training)
    d["fp2_features"], d["fp2_xyz"] = f, d["sa2_xyz"]
    d["fp2_inds"] = d["sa1_inds"][:, :d["fp2_xyz"].shape[1]]
    vote_xyz, vote_features = voting_module(W, vgen, d["fp2_xyz"], f, training)
    norm = np.sqrt((vote_features.astype(np.float64) ** 2).sum(axis=1, keepdims=True))
    vote_features = (vote_features / norm).astype(np.float32)
    d["vote_xyz"], d["vote_features"] = vote_xyz, vote_features
    agg_xyz, agg_f, agg_inds = orc.sa_module_votes(vote_xyz, vote_features, mlp_layers(W, vote_agg + "mlp_module."),
                                                   num_proposal, 0.3, 16, training, normalize_xyz=True)
    d["aggregated_vote_xyz"], d["aggregated_vote_inds"] = agg_xyz, agg_inds
    d["aggregated_vote_features"] = np.ascontiguousarray(agg_f.transpose(0, 2, 1))
    return d

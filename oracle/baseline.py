"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU baseline for bench.py's `cpu_baseline` leg (kind "port").

One full TRAINING step (forward + the reference's loss + backward + AdamW) of the grounding path on the host CPU:
  * geometry (FPS, ball query, three_nn) by the C/OpenMP restatement of the reference's kernels (pointnet2_oracle.c);
  * grouping / interpolation as torch index ops, every dense layer (SharedMLP 1x1 convs + BatchNorm, voting, ROI heads,
    relation / match attention, contrast heads) and the loss (3dvlp_amd.losses, impl="torch") through PyTorch-CPU
    autograd — the same nn.Modules and parameters as the GPU path, on their explicit pure-torch branches.
The reference's own ops have no CPU implementation at all (sampling.cpp:39 "CPU not supported"), so this port is what
"the reference's CPU path" can mean on this box.  It is a reported baseline, not a target.
"""
import importlib
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import oracle as orc


def threads_used():
    """Threads the C/OpenMP part runs on."""
    return int(orc.lib().orc_num_threads())


def _np(t):
    return t.detach().numpy()


def _sa_cpu(sa, xyz, features):
    """PointnetSAModuleVotes.forward (pointnet2_modules.py:210-272) on CPU tensors: oracle geometry + torch dense."""
    B, N, _ = xyz.shape
    x = np.ascontiguousarray(_np(xyz), np.float32)
    inds = orc.furthest_point_sampling(x, sa.npoint)
    new_x = np.stack([x[b, inds[b]] for b in range(B)])
    idx = torch.from_numpy(orc.ball_query(new_x, x, sa.radius, sa.nsample).astype(np.int64))  # (B,M,S)
    M, S = idx.shape[1:]
    ii = torch.from_numpy(inds.astype(np.int64))
    new_xyz = torch.gather(xyz, 1, ii[..., None].expand(-1, -1, 3))                            # autograd to xyz
    flat = idx.reshape(B, 1, M * S)
    gx = torch.gather(xyz.transpose(1, 2), 2, flat.expand(-1, 3, -1)).reshape(B, 3, M, S)
    gx = (gx - new_xyz.transpose(1, 2).unsqueeze(-1)) / sa.radius
    gf = torch.gather(features, 2, flat.expand(-1, features.shape[1], -1)).reshape(B, -1, M, S)
    y = sa.mlp_module(torch.cat([gx, gf], 1))
    return new_xyz, F.max_pool2d(y, kernel_size=[1, S]).squeeze(-1), ii


def _fp_cpu(fp, unknown, known, unknow_feats, known_feats):
    """PointnetFPModule.forward (pointnet2_modules.py:371-416)."""
    d2, idx = orc.three_nn(np.ascontiguousarray(_np(unknown), np.float32), np.ascontiguousarray(_np(known), np.float32))
    dist = torch.from_numpy(np.sqrt(d2)).to(known_feats.dtype)
    recip = 1.0 / (dist + 1e-8)
    w = recip / recip.sum(2, keepdim=True)
    idx = torch.from_numpy(idx.astype(np.int64))
    B, n, _ = idx.shape
    C = known_feats.shape[1]
    g = torch.gather(known_feats, 2, idx.reshape(B, 1, n * 3).expand(-1, C, -1)).reshape(B, C, n, 3)
    interp = (g * w.unsqueeze(1)).sum(-1)
    return fp.mlp(torch.cat([interp, unknow_feats], 1).unsqueeze(-1)).squeeze(-1)


class CpuStep:
    """The whole step on the CPU.  `net` is a 3dvlp_amd.grounding_step.GroundingNet built on the CPU."""

    def __init__(self, seed=0, lr=1e-3, dtype=torch.float32, use_answer=False, num_answers=0, use_caption=False,
                 caption_kwargs=None):
        """dtype=torch.float64: the dense layers, the loss and AdamW in double precision (geometry stays the fp32 C
        restatement: indices are defined by fp32 bits) — the yardstick of tests/test_step_parity.py."""
        self.gs = importlib.import_module("3dvlp_amd.grounding_step")
        self.losses = importlib.import_module("3dvlp_amd.losses")
        tr = importlib.import_module("3dvlp_amd.transformer")
        torch.manual_seed(seed)
        self.net = self.gs.GroundingNet(use_answer=use_answer, num_answers=num_answers, use_caption=use_caption,
                                        caption_kwargs=caption_kwargs).train().to(dtype)
        self.use_answer, self.use_caption = use_answer, use_caption
        self.dtype = dtype
        for m in self.net.modules():
            if isinstance(m, tr.ScaledDotProductAttention):
                m.impl = "torch"      # explicit unfused attention: the CPU branch of the module
        self.opt = torch.optim.AdamW(self.net.parameters(), lr=lr, weight_decay=1e-5)

    def trunk(self, batch):
        """Backbone + voting + L2 norm only (backbone_module.py:76-135, voting_module.py:33-60, jointnet.py:148-149):
        -> fp2_features (B,C,S), vote_xyz (B,S,3), vote_features (B,C,S).  Everything here is a continuous function of the
        dense layers' arithmetic (the geometry depends on the input coordinates alone), which the whole step is not."""
        net = self.net
        pc = batch["point_clouds"]
        xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
        bb = net.backbone_net
        lv = []
        for sa in (bb.sa1, bb.sa2, bb.sa3, bb.sa4):
            xyz, feats, inds = _sa_cpu(sa, xyz, feats)
            lv.append((xyz, feats, inds))
        f1 = _fp_cpu(bb.fp1, lv[2][0], lv[3][0], lv[2][1], lv[3][1])
        f = _fp_cpu(bb.fp2, lv[1][0], lv[2][0], lv[1][1], f1)
        vx, vf = net.vgen(lv[1][0], f)
        return f, vx, vf.div(torch.norm(vf, p=2, dim=1).unsqueeze(1))

    def forward_loss(self, batch):
        net = self.net
        d = dict(batch)
        d["epoch"] = 50
        pc = d["point_clouds"]
        xyz, feats = pc[..., :3].contiguous(), pc[..., 3:].transpose(1, 2).contiguous()
        bb = net.backbone_net
        lv = []
        for sa in (bb.sa1, bb.sa2, bb.sa3, bb.sa4):
            xyz, feats, inds = _sa_cpu(sa, xyz, feats)
            lv.append((xyz, feats, inds))
        f1 = _fp_cpu(bb.fp1, lv[2][0], lv[3][0], lv[2][1], lv[3][1])
        f = _fp_cpu(bb.fp2, lv[1][0], lv[2][0], lv[1][1], f1)
        if getattr(self, "keep", None) is not None:  # tests: intermediate activations with their gradients retained
            for name, t in (("sa1_features", lv[0][1]), ("sa2_features", lv[1][1]), ("sa3_features", lv[2][1]),
                            ("sa4_features", lv[3][1]), ("fp1_features", f1), ("fp2_features", f)):
                t.retain_grad()
                self.keep[name] = t
        d["seed_inds"], d["seed_xyz"], d["seed_features"] = lv[0][2][:, :lv[1][0].shape[1]].int(), lv[1][0], f
        vx, vf = net.vgen(lv[1][0], f)
        vf = vf.div(torch.norm(vf, p=2, dim=1).unsqueeze(1))
        d["vote_xyz"], d["vote_features"] = vx, vf
        if getattr(self, "keep", None) is not None:
            for name, t in (("vote_xyz", vx), ("vote_features", vf)):
                t.retain_grad()
                self.keep[name] = t
        prop = net.proposal
        ax, af, ai = _sa_cpu(prop.vote_aggregation, vx, vf)
        d["aggregated_vote_xyz"], d["aggregated_vote_inds"] = ax, ai.int()
        d["aggregated_vote_features"] = af.permute(0, 2, 1).contiguous()
        if getattr(self, "keep", None) is not None:
            for name, t in (("aggregated_vote_xyz", ax), ("aggregated_vote_features", d["aggregated_vote_features"])):
                t.retain_grad()
                self.keep[name] = t
        d = prop.decode_scores(prop.proposal(af, d))
        d = net.match(net.relation(d))
        d = net.constrast(d)
        cap_loss = None
        if self.use_caption:
            # the caption head by the op-by-op restatement oracle/captioner.py (eval-mode arithmetic: dropout and the MLM
            # corruption off — the caller builds the GPU step the same way) on the head's parameters under the reference's
            # key names (caption.py's reference_view: autograd reaches the module's own parameters through it), and
            # loss_captioning.py:25-48; every proposal is a good box (d2 > -1, transformer_captioner.py:476-480)
            from . import captioner as ocap
            cap = net.caption
            sd = cap.reference_view({n: p for n, p in cap.named_parameters()})
            for k, v in cap.state_dict().items():
                sd.setdefault(k, v)
            lang_cap, idx = ocap.forward_train(sd, d, N=cap.N, h=cap.h, early_guide=cap.early_guide)
            good = torch.ones(lang_cap.shape[0], dtype=lang_cap.dtype)
            cap_loss = ocap.cap_loss(lang_cap, d["input_ids"], good)
            d["match_idx"], d["lang_cap"] = idx, lang_cap
        args = None
        if self.use_answer:
            d = net.answer(d)
            args = type("Args", (self.losses._Args,), {"use_answer": True})
        self.losses.get_joint_loss(args, d, config=net.dataset_config, impl="torch")
        if cap_loss is not None:                       # loss_joint.py:222-223
            d["cap_loss"] = cap_loss
            d["loss"] = d["loss"] + cap_loss
        self.last = d
        return d["loss"]

    def step(self, batch):
        self.opt.zero_grad(set_to_none=True)
        loss = self.forward_loss(batch)
        loss.backward()
        self.opt.step()
        return float(loss.detach())


def to_torch(batch_np, scenes, dtype=torch.float32):
    L = batch_np["lang_fea"].shape[0] // batch_np["point_clouds"].shape[0]
    out = {}
    for k, v in batch_np.items():
        per_sentence = k in ("lang_fea", "lang_emb", "answer_cat_scores", "answer_cat")
        t = torch.from_numpy(np.ascontiguousarray(v[:scenes * L] if per_sentence else v[:scenes]))
        out[k] = t.to(dtype) if t.is_floating_point() else t
    out["istrain"] = [1]
    out["random"] = torch.tensor(0.75)
    return out


def timed_step(batch_np, scenes, threads, reps=1):
    """Seconds per full step (fwd + loss + bwd + AdamW) on `scenes` scenes with `threads` host threads."""
    torch.set_num_threads(threads)
    orc.lib().orc_set_num_threads(int(threads))
    step = CpuStep()
    batch = to_torch(batch_np, scenes)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        step.step(batch)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def cpu_baseline(batch_np, scenes=1):
    """-> the `cpu_baseline` object of bench.py's JSON line: all host cores, plus a single-thread figure."""
    # the threads this process may really use: its affinity mask, capped at the GPU box's per-GPU CPU share (16) —
    # os.cpu_count() reports every core of the host (256 on the MI355X boxes) and oversubscribing them made the
    # all-cores run 100x SLOWER than one thread
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("VLP3D_CPU_BASELINE_THREADS", 16))))
    t_all = timed_step(batch_np, scenes, cores, reps=2)  # best of two: the first call pays one-time thread-pool start-up
    t_one = timed_step(batch_np, scenes, 1)
    torch.set_num_threads(cores)
    orc.lib().orc_set_num_threads(cores)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(scenes / t_all, 4), "unit": "scenes/s", "cores": cores, "kind": "port",
            "single_thread_value": round(scenes / t_one, 4), "cpu_model": model,
            "sample": f"{scenes} scene(s) of the same workload (40k pts, 256 proposals, 8 sentences) as ONE B = {scenes} training "
                      f"step (fwd + reference loss + bwd + AdamW; BatchNorm statistics over that one scene — not the B = 8 step "
                      f"of the GPU line, whose per-scene cost it approximates): C/OpenMP geometry + PyTorch-CPU autograd; "
                      f"{t_all:.2f} s on {cores} threads, {t_one:.2f} s on 1 thread"}

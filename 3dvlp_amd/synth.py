"""Deterministic synthetic ScanRefer-like scenes — the bench/test input generator.

A scene is a 8 m x 8 m x 3 m room sampled ON SURFACES (40 % floor, 20 % walls, 40 % faces of 12
axis-aligned boxes resting on the floor) with 5 mm jitter, shifted so that no point falls inside the
FPS skip ball |p|^2 <= 1e-3 (sampling_gpu.cu:106).  Features: 132 channels (multiview 128 + normal 3 +
height 1 in the reference: scripts/joint_scripts/train_3dvlp.py:82-83); GT boxes / votes follow the
VoteNet label layout the reference's losses read (lib/loss_helper/loss_detection.py:24-110).
Language is synthetic too: BERT is frozen and out of scope, so `lang_fea` (B*L,50,128) is N(0,1).
numpy only; nothing here touches the GPU.
"""
import numpy as np

ROOM = np.array([8.0, 8.0, 3.0], np.float32)
OFFSET = np.array([0.5, 0.5, 0.1], np.float32)
NUM_BOXES = 12
MAX_NUM_OBJ = 128
NUM_CLASS = 18
NUM_FEATURES = 132


def mean_size_arr():
    """18 synthetic per-class mean box sizes (stands in for scannet_reference_means.npz)."""
    rng = np.random.default_rng(12345)
    return rng.uniform(0.4, 1.4, size=(NUM_CLASS, 3)).astype(np.float32)


def make_scene(seed, num_points=40000, skip_points=0):
    rng = np.random.default_rng(seed)
    n_floor = int(0.4 * num_points)
    n_wall = int(0.2 * num_points)
    n_box = num_points - n_floor - n_wall

    centers = np.zeros((NUM_BOXES, 3), np.float32)
    sizes = rng.uniform(0.3, 1.5, size=(NUM_BOXES, 3)).astype(np.float32)
    centers[:, :2] = rng.uniform(1.0, 7.0, size=(NUM_BOXES, 2))
    centers[:, 2] = sizes[:, 2] / 2

    pts = [np.concatenate([rng.uniform(0, 1, (n_floor, 2)) * ROOM[:2], np.zeros((n_floor, 1))], 1)]
    inst = [np.full(n_floor, -1)]
    w = rng.integers(0, 4, n_wall)
    u = rng.uniform(0, 1, (n_wall, 2))
    wall = np.zeros((n_wall, 3))
    wall[:, 2] = u[:, 1] * ROOM[2]
    along = u[:, 0] * ROOM[0]
    wall[:, 0] = np.where(w == 0, 0.0, np.where(w == 1, ROOM[0], along))
    wall[:, 1] = np.where(w == 2, 0.0, np.where(w == 3, ROOM[1], along))
    wall[w < 2, 1] = along[w < 2]
    pts.append(wall)
    inst.append(np.full(n_wall, -1))

    # box faces: 4 sides + top, chosen by area
    which = rng.integers(0, NUM_BOXES, n_box)
    s, c = sizes[which], centers[which]
    areas = np.stack([s[:, 1] * s[:, 2], s[:, 1] * s[:, 2], s[:, 0] * s[:, 2], s[:, 0] * s[:, 2], s[:, 0] * s[:, 1]], 1)
    cdf = np.cumsum(areas / areas.sum(1, keepdims=True), 1)
    face = (rng.uniform(0, 1, (n_box, 1)) > cdf).sum(1).clip(0, 4)
    uv = rng.uniform(-0.5, 0.5, (n_box, 3))
    uv[face == 0, 0] = -0.5
    uv[face == 1, 0] = 0.5
    uv[face == 2, 1] = -0.5
    uv[face == 3, 1] = 0.5
    uv[face == 4, 2] = 0.5
    pts.append(c + uv * s)
    inst.append(which)

    xyz = np.concatenate(pts, 0) + rng.normal(0, 0.005, (num_points, 3))
    inst = np.concatenate(inst, 0)
    perm = rng.permutation(num_points)
    xyz, inst = (xyz[perm] + OFFSET).astype(np.float32), inst[perm]
    centers = centers + OFFSET
    if skip_points:  # variant scene that exercises the FPS skip rule
        xyz[rng.choice(num_points, skip_points, replace=False)] = rng.uniform(-0.01, 0.01, (skip_points, 3))

    feats = rng.normal(0, 1, (num_points, NUM_FEATURES)).astype(np.float32)
    feats[:, -1] = xyz[:, 2] - np.percentile(xyz[:, 2], 1)  # height channel (lib/joint/dataset.py:603-607)

    # VoteNet labels: each object point votes (3 identical copies) for its box centre
    vote_label = np.zeros((num_points, 9), np.float32)
    vote_mask = (inst >= 0).astype(np.int64)
    obj = inst >= 0
    vote_label[obj] = np.tile(centers[inst[obj]] - xyz[obj], (1, 3))

    means = mean_size_arr()
    size_class = np.argmin(((sizes[:, None, :] - means[None]) ** 2).sum(-1), 1)
    center_label = np.zeros((MAX_NUM_OBJ, 3), np.float32)
    center_label[:NUM_BOXES] = centers
    box_mask = np.zeros(MAX_NUM_OBJ, np.float32)
    box_mask[:NUM_BOXES] = 1
    size_residual = (sizes - means[size_class]).astype(np.float32)
    # per-object GT arrays padded to MAX_NUM_OBJ, as lib/joint/dataset.py:826-840 hands them to the losses
    # (ScanNet: axis-aligned boxes, heading class / residual 0; semantic class == size class)
    pad = lambda a: np.concatenate([a, np.zeros((MAX_NUM_OBJ - NUM_BOXES,) + a.shape[1:], a.dtype)], 0)
    # instance ids per point as the loader sees them (lib/joint/dataset.py:592): boxes 0..11, everything else one
    # un-annotated instance (id NUM_BOXES) — the augmentation recomputes the votes from these (dataset.py:653-679)
    instance_labels = np.where(inst >= 0, inst, NUM_BOXES).astype(np.int32)
    return dict(xyz=xyz, features=feats, vote_label=vote_label, vote_label_mask=vote_mask, center_label=center_label,
                instance_labels=instance_labels,
                box_label_mask=box_mask, box_centers=centers.astype(np.float32), box_sizes=sizes,
                size_class=size_class, size_residual=size_residual,
                heading_class_label=np.zeros(MAX_NUM_OBJ, np.int64), heading_residual_label=np.zeros(MAX_NUM_OBJ, np.float32),
                size_class_label=pad(size_class.astype(np.int64)), size_residual_label=pad(size_residual),
                sem_cls_label=pad(size_class.astype(np.int64)))


def make_batch(first_scene, batch_size, num_points=40000, lang_num_max=8, seed_base=1000, num_answers=0, instances=False,
               caption_tokens=0):
    """Batch dict of numpy arrays with the keys the grounding step reads (jointnet.py / loss_joint.py).  num_answers > 0
    adds the ScanQA targets of the joint QA + grounding task: `answer_cat_scores` (B*L, num_answers) soft scores (1-3
    annotated answers per question, VQA-style min(1, 0.3 count)) and `answer_cat` (B*L) the first of them.
    caption_tokens = T > 0: `input_ids` becomes (B, L, T) BERT-like token ids for the caption head (BASELINE cfg4): [CLS] = 101,
    8 .. T - 2 random word ids in [1000, 30000), [SEP] = 102, then pads (0) — the description tokens of lib/joint/dataset.py.
    instances=True adds what the training-time augmentation needs (input_pipeline.augment_on_device): `instance_labels`
    (B,N) int32, `instance_valid` (B, NUM_BOXES + 1) uint8 and `box_sizes` (B, MAX_NUM_OBJ, 3)."""
    scenes = [make_scene(seed_base + first_scene + i, num_points) for i in range(batch_size)]
    rng = np.random.default_rng(777 + first_scene)
    pc = np.stack([np.concatenate([s["xyz"], s["features"]], 1) for s in scenes])
    L = lang_num_max
    target = rng.integers(0, NUM_BOXES, (batch_size, L))
    out = dict(
        point_clouds=pc.astype(np.float32),
        vote_label=np.stack([s["vote_label"] for s in scenes]),
        vote_label_mask=np.stack([s["vote_label_mask"] for s in scenes]),
        center_label=np.stack([s["center_label"] for s in scenes]),
        box_label_mask=np.stack([s["box_label_mask"] for s in scenes]),
        heading_class_label=np.stack([s["heading_class_label"] for s in scenes]),
        heading_residual_label=np.stack([s["heading_residual_label"] for s in scenes]),
        size_class_label=np.stack([s["size_class_label"] for s in scenes]),
        size_residual_label=np.stack([s["size_residual_label"] for s in scenes]),
        sem_cls_label=np.stack([s["sem_cls_label"] for s in scenes]),
        lang_fea=rng.normal(0, 1, (batch_size * L, 50, 128)).astype(np.float32),
        lang_num=np.full(batch_size, L, np.int64),
        input_ids=np.zeros((batch_size, L, 50), np.int64),
        ref_box_label_list=target.astype(np.int64),
        ref_center_label_list=np.stack([s["box_centers"][t] for s, t in zip(scenes, target)]),
        ref_size_class_label_list=np.stack([s["size_class"][t] for s, t in zip(scenes, target)]).astype(np.int64),
        ref_size_residual_label_list=np.stack([s["size_residual"][t] for s, t in zip(scenes, target)]),
        ref_heading_class_label_list=np.zeros((batch_size, L), np.int64),
        ref_heading_residual_label_list=np.zeros((batch_size, L), np.float32),
    )
    out["lang_emb"] = out["lang_fea"][:, 0].copy()
    if instances:
        out["instance_labels"] = np.stack([s["instance_labels"] for s in scenes])
        valid = np.ones((batch_size, NUM_BOXES + 1), np.uint8)
        valid[:, NUM_BOXES] = 0
        out["instance_valid"] = valid
        sizes = np.zeros((batch_size, MAX_NUM_OBJ, 3), np.float32)
        sizes[:, :NUM_BOXES] = np.stack([s["box_sizes"] for s in scenes])
        out["box_sizes"] = sizes
    if num_answers:
        sc = np.zeros((batch_size * L, num_answers), np.float32)
        first = np.zeros(batch_size * L, np.int64)
        for q in range(batch_size * L):
            picks = rng.choice(num_answers, size=int(rng.integers(1, 4)), replace=False)
            sc[q, picks] = np.minimum(1.0, 0.3 * rng.integers(1, 5, size=len(picks))).astype(np.float32)
            first[q] = picks[0]
        out["answer_cat_scores"], out["answer_cat"] = sc, first
    if caption_tokens:
        T = int(caption_tokens)
        r2 = np.random.default_rng(4242 + first_scene)    # its own stream: the other arrays do not depend on this switch
        ids = np.zeros((batch_size, L, T), np.int64)
        ids[..., 0] = 101
        for b in range(batch_size):
            for l in range(L):
                n = int(r2.integers(min(8, T - 2), T - 1))
                ids[b, l, 1:1 + n - 1] = r2.integers(1000, 30000, n - 1)
                ids[b, l, n] = 102
        out["input_ids"] = ids
    return out

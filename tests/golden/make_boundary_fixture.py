"""Boundary proof: the reference's OWN caller modules, imported unchanged on top of this repository's drop-in modules.

Run in the build container only (needs /root/reference; nothing of it is copied):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_boundary_fixture.py

INTEGRATION.md §2 claims that after aliasing `lib.pointnet2.pointnet2_utils`, `lib.pointnet2.pointnet2_modules`,
`lib.pointnet2.pytorch_utils` (and the bare `pointnet2_*` / `pytorch_utils` names) to `3dvlp_amd.*`, the reference's
callers import and construct unchanged.  This script does exactly that for
    models/base_module/backbone_module.py        (Pointnet2Backbone)
    models/base_module/voting_module.py          (VotingModule)
    models/proposal_module/ROI_heads/roi_heads.py (StandardROIHeads; proposal_module_fcos.py needs easydict -> see below)
    models/proposal_module/relation_module.py    (RelationModule)
    models/refnet/match_module.py                (MatchModule)
with the arguments models/jointnet/jointnet.py:62-100 passes, and records every state_dict key, shape and dtype —
tests/test_boundary.py then requires 3dvlp_amd.detection / grounding / GroundingNet to expose the same keys, i.e. a
reference checkpoint loads into the drop-in and vice versa.  The fixture is DATA (names and shapes), no source text.
"""
import importlib
import json
import os
import sys

import numpy as np

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.dont_write_bytecode = True
sys.path[:0] = [ROOT]
# 1. the aliases of INTEGRATION.md §2 — BEFORE any reference module is imported
importlib.import_module("3dvlp_amd")
ALIASES = {
    "pointnet2_utils": "3dvlp_amd.pointnet2_utils", "lib.pointnet2.pointnet2_utils": "3dvlp_amd.pointnet2_utils",
    "pointnet2_modules": "3dvlp_amd.pointnet2_modules", "lib.pointnet2.pointnet2_modules": "3dvlp_amd.pointnet2_modules",
    "pytorch_utils": "3dvlp_amd.pytorch_utils", "lib.pointnet2.pytorch_utils": "3dvlp_amd.pytorch_utils",
}
for name, target in ALIASES.items():
    sys.modules[name] = importlib.import_module(target)
# 2. the reference tree (its lib/pointnet2 directory is NOT put on the path: the aliases must carry the imports)
sys.path.append(REF)

from models.base_module.backbone_module import Pointnet2Backbone  # noqa: E402
from models.base_module.voting_module import VotingModule  # noqa: E402
from models.proposal_module.ROI_heads.roi_heads import StandardROIHeads  # noqa: E402
from models.proposal_module.relation_module import RelationModule  # noqa: E402
from models.refnet.match_module import MatchModule  # noqa: E402


def describe(module):
    return {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in module.state_dict().items()}


def main():
    mean_size_arr = np.ones((18, 3), np.float32)
    # jointnet.py:62-100 with train_3dvlp.py's arguments: num_class 18, 1 heading bin, 18 size clusters, 132 input
    # features (multiview + normal + height), 256 proposals, vote_factor 1, sampling "vote_fps"
    mods = {
        "backbone_net": Pointnet2Backbone(input_feature_dim=132),
        "vgen": VotingModule(1, 256),
        # models/proposal_module/proposal_module_fcos.py itself cannot be imported here (its `from data.scannet.
        # model_util_scannet import ...` needs easydict: an ordinary ImportError); its two sub-modules are
        # pinned separately: the ROI heads from the reference's class, the vote aggregation as constructed at :36-43
        "proposal.proposal": StandardROIHeads(num_heading_bin=1, num_class=18, seed_feat_dim=256, use_kl_loss=False),
        "proposal.vote_aggregation": sys.modules["lib.pointnet2.pointnet2_modules"].PointnetSAModuleVotes(
            npoint=256, radius=0.3, nsample=16, mlp=[256, 128, 128, 128], use_xyz=True, normalize_xyz=True),
        "relation": RelationModule(num_proposals=256, det_channel=128),
        "match": MatchModule(num_proposals=256, lang_size=256, det_channel=128, use_lang_emb=False, use_pc_encoder=False,
                             use_match_con_loss=False, use_reg_head=False),
    }
    # which classes did the reference's modules actually instantiate?  (must be this repository's)
    sa1 = mods["backbone_net"].sa1
    origin = {"sa1": type(sa1).__module__, "sa1.mlp_module": type(sa1.mlp_module).__module__,
              "fp1": type(mods["backbone_net"].fp1).__module__,
              "fp1.mlp": type(mods["backbone_net"].fp1.mlp).__module__}
    assert all(v.startswith("3dvlp_amd.") for v in origin.values()), origin
    out = {"origin": origin, "state_dict": {k: describe(m) for k, m in mods.items()}}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "boundary_state_dict.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path, sum(len(v) for v in out["state_dict"].values()), "keys;", origin)


if __name__ == "__main__":
    main()

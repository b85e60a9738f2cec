"""Drop-in boundary, checked against the reference's own caller modules (CPU).

tests/golden/boundary_state_dict.json was written by tests/golden/make_boundary_fixture.py, which imports the
reference's backbone_module / voting_module / relation_module / match_module / roi_heads UNCHANGED on top of this
package's modules (sys.modules aliases of INTEGRATION.md §2), constructs them with jointnet.py:62-100's arguments and
records every state_dict entry.  Here: this package's own GroundingNet exposes exactly those keys, shapes and dtypes
— a reference checkpoint loads into it (and one of ours into the reference's modules).
"""
import importlib
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _fixture():
    return json.load(open(os.path.join(HERE, "golden", "boundary_state_dict.json")))


def test_reference_callers_instantiated_this_packages_classes():
    origin = _fixture()["origin"]
    assert origin["sa1"] == "3dvlp_amd.pointnet2_modules" and origin["fp1"] == "3dvlp_amd.pointnet2_modules"
    assert origin["sa1.mlp_module"] == "3dvlp_amd.pytorch_utils" and origin["fp1.mlp"] == "3dvlp_amd.pytorch_utils"


def test_grounding_net_state_dict_equals_reference_modules():
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    net = gs.GroundingNet()
    mine = {k: (list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in net.state_dict().items()}
    ref = _fixture()["state_dict"]
    seen = set()
    for prefix, entries in ref.items():
        assert entries, prefix
        for k, (shape, dtype) in entries.items():
            full = f"{prefix}.{k}"
            assert full in mine, full
            assert mine[full] == (shape, dtype), (full, mine[full], shape, dtype)
            seen.add(full)
        # ... and nothing extra under that prefix (strict load both ways)
        extra = [k for k in mine if k.startswith(prefix + ".") and k not in seen]
        assert not extra, (prefix, extra[:5])
    # what remains are the modules the reference keeps elsewhere (ContrastModule: not importable here, pytorch3d)
    rest = {k.split(".")[0] for k in mine if k not in seen}
    assert rest <= {"constrast"}, rest


def test_reference_style_checkpoint_round_trip():
    """A state_dict with the reference's keys loads strictly; `lang.*` / `caption.*` entries of a full JointNet
    checkpoint are the only unexpected ones (strict=False)."""
    gs = importlib.import_module("3dvlp_amd.grounding_step")
    a, b = gs.GroundingNet(), gs.GroundingNet()
    sd = {k: torch.randn_like(v) if v.dtype.is_floating_point else v for k, v in a.state_dict().items()}
    b.load_state_dict(sd, strict=True)
    sd["lang.fc.weight"] = torch.zeros(3)
    res = b.load_state_dict(sd, strict=False)
    assert res.missing_keys == [] and res.unexpected_keys == ["lang.fc.weight"]


def test_prepared_weights_keys_and_pass_counter():
    """Host logic of row_mlp.PreparedWeights: entries are keyed by (address, shape) — two views of one address
    with different shapes do not evict each other — and every refresh advances `pass_id`, which a backward that kept
    K-major copies (row_chain) compares before it multiplies by them."""
    import importlib
    import torch
    rm = importlib.import_module("3dvlp_amd.row_mlp")
    prep = rm.PreparedWeights()
    w = torch.randn(64, 32)
    a, b = w.view(64, 32), w.view(32, 64)
    assert prep.lookup(a) is None and prep.lookup(b) is None          # first sight: registered, not served
    assert len(prep.entries) == 2 and len(prep.fresh) == 2
    assert prep.lookup(a) is None                                     # still not served inside the registering pass
    prep.fresh.clear()                                                # (what __enter__ does after the batched transpose)
    ta, tb = prep.lookup(a), prep.lookup(b)
    assert ta.shape == (32, 64) and tb.shape == (64, 32) and ta.data_ptr() != tb.data_ptr()
    p0 = prep.pass_id
    assert prep.current(p0)
    prep.pass_id += 1                                                 # a later forward pass refreshed the copies
    assert not prep.current(p0) and prep.current(p0 + 1)

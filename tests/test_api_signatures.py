"""CPU: the drop-in boundary of SURVEY.md section 8(b), checked mechanically.

tests/golden/api_signatures.npz holds ``inspect.signature`` of every class / method on the boundary, taken from THE
REFERENCE'S OWN modules (tests/golden/make_golden.py:gen_api).  Each vlfm_amd counterpart must accept the same
parameters, in the same order, of the same kind and with the same defaults, so that a reference caller -- positional or
keyword -- binds identically.  vlfm_amd may append optional parameters (``device=...``) after the reference's.
Deviations are listed here with their reason; anything else fails.
"""
import importlib
import inspect
import json

import pytest

from golden_util import load

REF = json.loads(str(load("api_signatures")["json"]))

# (qualified method, parameter) -> reason.  All are model constructors whose checkpoints do not exist offline.
RELAXED_DEFAULTS = {
    ("vlfm.vlm.yolov7.YOLOv7.__init__", "weights"): "required path to yolov7-e6e.pt -> optional (random-init / TorchScript via env)",
    ("vlfm.vlm.sam.MobileSAM.__init__", "sam_checkpoint"): "required path to mobile_sam.pt -> optional (random-init)",
    ("vlfm.vlm.grounding_dino.GroundingDINO.__init__", "config_path"): "groundingdino python-config path -> None (HF config)",
    ("vlfm.vlm.grounding_dino.GroundingDINO.__init__", "weights_path"): "checkpoint path -> None (HF model dir via env)",
    ("vlfm.vlm.grounding_dino.GroundingDINO.__init__", "device"): "torch.device('cuda') evaluated at import -> None = current HIP device",
}


def _mine(qualified):
    mod, cls, method = qualified.rsplit(".", 2)
    c = getattr(importlib.import_module(mod.replace("vlfm.", "vlfm_amd.", 1)), cls)
    return inspect.signature(getattr(c, method))


@pytest.mark.parametrize("qualified", sorted(REF))
def test_signature_binds_like_the_reference(qualified):
    want = REF[qualified]
    got = list(_mine(qualified).parameters.values())
    assert len(got) >= len(want), f"{qualified}: missing parameters {[w[0] for w in want[len(got):]]}"
    for (name, kind, default), p in zip(want, got):
        assert p.name == name and p.kind.name == kind, f"{qualified}: {p.name}/{p.kind.name} != {name}/{kind}"
        have = None if p.default is inspect.Parameter.empty else repr(p.default)
        if (qualified, name) in RELAXED_DEFAULTS:
            assert have is not None  # relaxed = became optional, never the other way round
        elif default == "<function>":
            assert callable(p.default)
        else:
            assert have == default, f"{qualified}({name}): default {have} != {default}"
    for p in got[len(want):]:  # our additions must not change how a reference call binds
        assert p.default is not inspect.Parameter.empty or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD), \
            f"{qualified}: extra parameter {p.name} without a default"


def test_boundary_attributes_exist():
    """Attributes the reference's callers read."""
    from vlfm_amd.mapping import BaseMap, ObstacleMap, ValueMap
    from vlfm_amd.vlm.detections import ObjectDetections

    for cls, names in ((ValueMap, ["update_map", "sort_waypoints", "reset", "update_agent_traj", "visualize"]),
                       (ObstacleMap, ["update_map", "reset", "update_agent_traj", "visualize", "radius_padding_color"]),
                       (BaseMap, ["_xy_to_px", "_px_to_xy"]),
                       (ObjectDetections, ["filter_by_conf", "filter_by_class", "num_detections", "to_json", "from_json",
                                           "annotated_frame"])):
        for n in names:
            assert hasattr(cls, n), f"{cls.__name__}.{n}"


def test_live_reference_signatures_equal_the_fixture():
    from oracle import ref_shim

    if not ref_shim.available():
        pytest.skip("/root/reference not present")
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden

    assert json.loads(str(make_golden.gen_api()["json"])) == REF

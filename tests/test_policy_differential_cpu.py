"""The frontier-selection rule (stick-to-last within 0.01, 0.5 m re-match, best non-cyclic otherwise) of
vlfm_amd/policy_step.py:FrontierSelector against the REFERENCE'S OWN method, run here: /root/reference/vlfm/policy/itm_policy.py:
BaseITMPolicy._get_best_frontier is imported through oracle/ref_shim.py (stand-ins for the absent third-party imports only) and driven
on random sequences of frontier sets -- frontiers that persist, move by less / more than 0.5 m, disappear and come back, values that
drift around the 0.01 hysteresis, ties, single frontiers.  Skipped where /root/reference does not exist (the GPU box); the recorded
episodes of tests/golden/policy_*.npz cover the same rule there."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/vlfm"), reason="needs the reference checkout")


def _reference_chooser():
    from oracle import ref_shim

    itm, _ = ref_shim.reference_policy()
    from vlfm.policy.utils.acyclic_enforcer import AcyclicEnforcer

    class Ref(itm.BaseITMPolicy):      # only the state _get_best_frontier touches (itm_policy.py:48-54, 59)
        def __init__(self):
            self._acyclic_enforcer = AcyclicEnforcer()
            self._last_value = float("-inf")
            self._last_frontier = np.zeros(2)
            self._observations_cache = {}
            self.sorted = None

        def _sort_frontiers_by_value(self, observations, frontiers):
            return self.sorted

    return Ref()


@pytest.mark.parametrize("seed", range(6))
def test_frontier_choice_equals_the_reference_method_on_random_sequences(seed, capsys):
    from vlfm_amd.policy_step import FrontierSelector

    rng = np.random.default_rng(seed)
    ref, ours = _reference_chooser(), FrontierSelector()
    pool = rng.uniform(-8, 8, (12, 2))
    stuck = 0
    for step in range(300):
        # the frontier set of this step: a subset of a slowly moving pool, sometimes jittered by less / more than the 0.5 m re-match
        pool += rng.normal(0, 0.02, pool.shape) * (rng.uniform(size=(12, 1)) < 0.3)
        if rng.uniform() < 0.05:
            pool[rng.integers(0, 12)] = rng.uniform(-8, 8, 2)
        keep = rng.uniform(size=12) < rng.uniform(0.2, 0.9)
        if not keep.any():
            keep[rng.integers(0, 12)] = True
        pts = pool[keep].copy()
        if rng.uniform() < 0.3:
            pts += rng.uniform(-0.45, 0.45, pts.shape) / np.sqrt(2)
        if rng.uniform() < 0.3 and step and not np.array_equal(ours.last_frontier, np.zeros(2)):
            pts[rng.integers(0, len(pts))] = ours.last_frontier          # the pursued frontier is listed again, exactly
        vals = rng.choice([0.1, 0.2, 0.21, 0.205, 0.3], len(pts)) + rng.choice([0.0, 0.004, -0.004, 0.011, -0.011], len(pts))
        order = np.argsort(-vals, kind="stable")
        sorted_pts, sorted_vals = pts[order], [float(v) for v in vals[order]]
        robot = rng.uniform(-8, 8, 2) if rng.uniform() < 0.2 else np.round(rng.uniform(-8, 8, 2), 1)
        ref.sorted = (sorted_pts.copy(), list(sorted_vals))
        ref._observations_cache = {"robot_xy": robot.copy()}
        want_pt, want_val = ref._get_best_frontier({}, pts.copy())
        got_pt, got_val = ours.choose(sorted_pts.copy(), list(sorted_vals), pts.copy(), robot.copy())
        assert np.array_equal(np.asarray(got_pt), np.asarray(want_pt)) and got_val == want_val, (seed, step)
        assert np.array_equal(ours.last_frontier, ref._last_frontier) and ours.last_value == ref._last_value
        stuck += "Sticking" in os.environ.get("DEBUG_INFO", "")
    capsys.readouterr()                                   # (the reference prints per decision)
    assert 20 < stuck < 290                               # both branches were exercised


@pytest.mark.parametrize("seed", range(4))
def test_detector_routing_and_filters_equal_the_reference_method(seed):
    """base_objectnav_policy.py:221-241 (COCO targets -> YOLOv7 at 0.8, others -> GroundingDINO at 0.4, mixed targets retry) through
    the reference's own method and its own ObjectDetections, against ITMPolicyV2Step._get_object_detections over this package's
    ObjectDetections: random targets (COCO, non-COCO, mixed with '|'), random detection lists around both thresholds."""
    import types

    import torch

    from oracle import ref_shim

    _, base = ref_shim.reference_policy()
    import importlib

    ref_det = importlib.import_module("vlfm.vlm.detections")
    from vlfm_amd.policy_step import ITMPolicyV2Step
    from vlfm_amd.vlm.coco_classes import COCO_CLASSES
    from vlfm_amd.vlm.detections import ObjectDetections

    rng = np.random.default_rng(40 + seed)
    names = ["chair", "bed", "potted plant", "toilet", "tv", "couch", "cabinet", "fireplace", "shoe rack", "dog"]

    def fake(kind, calls):
        class D:
            def predict(self, img, caption=None):
                n = int(rng.integers(0, 6))
                boxes = torch.tensor(rng.uniform(0.05, 0.45, (n, 4)), dtype=torch.float32)
                boxes[:, 2:] += boxes[:, :2]
                logits = torch.tensor(rng.choice([0.2, 0.39, 0.4, 0.41, 0.79, 0.8, 0.81, 0.95], n), dtype=torch.float32)
                phrases = [str(rng.choice(names)) for _ in range(n)]
                calls.append((kind, caption, boxes.clone(), logits.clone(), list(phrases)))
                return None
        return D()

    for trial in range(200):
        k = int(rng.integers(1, 4))
        target = "|".join(rng.choice(names, k, replace=False))
        load_yolo = bool(rng.uniform() < 0.85)
        script = []                                         # the detector outputs are drawn once and replayed to both sides
        det_rng_state = rng.bit_generator.state
        outs = {}
        for side, DetCls in (("ref", ref_det.ObjectDetections), ("ours", ObjectDetections)):
            rng.bit_generator.state = det_rng_state
            calls = []

            def mk(kind):
                d = fake(kind, calls)
                inner = d.predict

                def predict(img, caption=None, _inner=inner):
                    _inner(img, caption)
                    _, _, boxes, logits, phrases = calls[-1]
                    return DetCls(boxes, logits, phrases, image_source=img, fmt="xyxy")
                d.predict = predict
                return d
            ns = types.SimpleNamespace(_target_object=target, _load_yolo=load_yolo, _coco_object_detector=mk("yolo"),
                                       _object_detector=mk("gdino"), _non_coco_caption="chair . bed . dog .",
                                       _coco_threshold=0.8, _non_coco_threshold=0.4)
            img = np.zeros((48, 64, 3), np.uint8)
            fn = base.BaseObjectNavPolicy._get_object_detections if side == "ref" else ITMPolicyV2Step._get_object_detections
            det = fn(ns, img)
            outs[side] = (np.asarray(det.boxes), np.asarray(det.logits), list(det.phrases), [(c[0], c[1]) for c in calls])
        assert outs["ref"][3] == outs["ours"][3], (trial, target)            # same detectors asked, in the same order, same captions
        assert outs["ref"][2] == outs["ours"][2], (trial, target)
        assert np.array_equal(outs["ref"][0], outs["ours"][0]) and np.array_equal(outs["ref"][1], outs["ours"][1])
    assert any(n not in COCO_CLASSES for n in names) and any(n in COCO_CLASSES for n in names)


@pytest.mark.parametrize("seed", range(3))
def test_object_detections_container_equals_the_reference_class_on_random_inputs(seed):
    """vlfm_amd/vlm/detections.py beside the reference's ObjectDetections (vlfm/vlm/detections.py:15-126; its torchvision box_convert is
    the shim's stand-in) on random boxes / scores / phrases: construction in every box format, filter_by_conf (>=, incl. a score equal
    to the threshold), filter_by_class, the early return when everything is kept, to_json / from_json round trips, repr, zero detections."""
    import torch

    from oracle import ref_shim
    from vlfm_amd.vlm.detections import ObjectDetections as Ours

    Ref = ref_shim.reference_detections().ObjectDetections
    rng = np.random.default_rng(300 + seed)
    names = ["chair", "bed", "tv", "potted plant", "dining table", "couch"]
    for trial in range(120):
        n = int(rng.integers(0, 9))
        fmt = ["cxcywh", "xyxy"][trial % 2]          # the two formats the reference constructs with (grounding_dino.py:69, yolov7.py:108)
        boxes = torch.from_numpy(rng.uniform(0, 1, (n, 4)).astype(np.float32))
        logits = torch.from_numpy(np.round(rng.uniform(0, 1, n), 2).astype(np.float32))
        phrases = [names[int(i)] for i in rng.integers(0, len(names), n)]
        a, b = Ours(boxes.clone(), logits.clone(), list(phrases), None, fmt=fmt), Ref(boxes.clone(), logits.clone(), list(phrases), None, fmt=fmt)

        def same():
            assert torch.equal(a.boxes, b.boxes) and torch.equal(a.logits, b.logits) and a.phrases == b.phrases
            assert a.num_detections == b.num_detections and repr(a) == repr(b) and a.to_json() == b.to_json()

        same()
        thr = float(logits[int(rng.integers(0, n))]) if n and trial % 2 else float(np.round(rng.uniform(0, 1), 2))
        a.filter_by_conf(thr); b.filter_by_conf(thr)
        same()
        keep = [names[int(i)] for i in rng.integers(0, len(names), int(rng.integers(0, 4)))]
        a.filter_by_class(keep); b.filter_by_class(keep)
        same()
        ja, jb = Ours.from_json(a.to_json()), Ref.from_json(b.to_json())
        assert torch.equal(ja.boxes.reshape(-1, 4) if ja.boxes.numel() else ja.boxes, jb.boxes.reshape(-1, 4) if jb.boxes.numel() else jb.boxes)
        assert ja.boxes.shape == jb.boxes.shape and ja.boxes.dtype == jb.boxes.dtype and torch.equal(ja.logits, jb.logits) and ja.phrases == jb.phrases


def test_geometry_helpers_equal_the_reference_functions_on_random_inputs():
    """The host-side helpers the decision path and the object map restate (vlfm_amd/policy_step.py, vlfm_amd/mapping/object_point_cloud_map.py)
    beside vlfm/utils/geometry_utils.py itself: xyz_yaw_to_tf_matrix, get_fov, rho_theta, closest_point_within_threshold, extract_yaw,
    within_fov_cone -- bit for bit on random inputs, incl. headings at +-pi, goals behind the agent, empty point sets, points on the cone's edge."""
    from oracle import ref_shim
    from vlfm_amd import policy_step as ps
    from vlfm_amd.mapping import object_point_cloud_map as opm

    geo = ref_shim.reference_modules()[2]
    rng = np.random.default_rng(12)
    for trial in range(400):
        xyz = rng.uniform(-20, 20, 3)
        yaw = float(rng.choice([rng.uniform(-np.pi, np.pi), np.pi, -np.pi, 0.0, np.pi / 2]))
        tf_a, tf_b = ps.xyz_yaw_to_tf_matrix(xyz, yaw), geo.xyz_yaw_to_tf_matrix(xyz, yaw)
        assert tf_a.dtype == tf_b.dtype and np.array_equal(tf_a, tf_b)
        assert opm.extract_yaw(tf_a) == geo.extract_yaw(tf_b)
        f, n = float(rng.uniform(50, 900)), int(rng.integers(8, 2000))
        assert ps.get_fov(f, n) == geo.get_fov(f, n)
        pos, goal = rng.uniform(-10, 10, 2), rng.uniform(-10, 10, 2)
        if trial % 7 == 0:
            goal = pos.copy()                                   # standing on the goal
        ra, rb = ps.rho_theta(pos, yaw, goal), geo.rho_theta(pos, yaw, goal)
        assert np.array_equal(np.asarray(ra, np.float64), np.asarray(rb, np.float64)), (trial, ra, rb)
        pts = rng.uniform(-5, 5, (int(rng.integers(0, 12)), 2))
        thr = float(rng.choice([0.5, 0.05, 3.0]))
        if len(pts):
            assert ps.closest_point_within_threshold(pts, pos / 4, thr) == geo.closest_point_within_threshold(pts, pos / 4, thr)
        cloud = np.concatenate([rng.uniform(-6, 6, (int(rng.integers(0, 40)), 3)), rng.uniform(size=(0, 3))])
        cloud = np.concatenate([cloud, rng.integers(0, 2, (len(cloud), 1)).astype(np.float64)], axis=1) if trial % 2 else cloud
        ang, fov, rng_m = yaw, float(rng.uniform(0.3, 2.5)), float(rng.uniform(0.5, 6))
        wa = opm.within_fov_cone(xyz / 4, ang, fov, rng_m, cloud)      # (the caller passes the 3-D camera position, object_point_cloud_map.py:99)
        wb = geo.within_fov_cone(xyz / 4, ang, fov, rng_m, cloud)
        assert wa.shape == wb.shape and np.array_equal(wa, wb), trial

"""CPU: the host-side decision path (vlfm_amd/policy_step.py).

* hand-derived cases for the frontier selection rule, the goal hand-over and the geometry helpers;
* when /root/reference is present: ITMPolicyV2Step over THE REFERENCE'S OWN map classes (imported through
  oracle/ref_shim.py) must reproduce, step by step, what the reference's own ITMPolicyV2 did on the same scripted
  episodes (tests/golden/policy_*.npz) -- this isolates the control flow from the GPU maps, which
  tests/test_policy_step_gpu.py then swaps in.
"""
import numpy as np
import pytest

from golden_util import POLICY_CASES, dense, load, replay_policy_episode, sha, unpack_plane
from vlfm_amd.policy_step import (AcyclicEnforcer, FrontierSelector, closest_point_within_threshold, get_fov,
                                  habitat_objectgoal_name, rho_theta, xyz_yaw_to_tf_matrix)


def test_geometry_helpers_known_answers():
    rho, theta = rho_theta(np.array([1.0, 1.0]), np.pi / 2, np.array([1.0, 3.0]))   # goal straight ahead of a robot facing +y
    assert abs(rho - 2.0) < 1e-12 and abs(theta) < 1e-12
    rho, theta = rho_theta(np.array([0.0, 0.0]), 0.0, np.array([0.0, 2.0]))          # goal to the left
    assert abs(rho - 2.0) < 1e-12 and abs(theta - np.pi / 2) < 1e-12
    assert abs(get_fov(320.0, 640) - np.pi / 2) < 1e-12
    tf = xyz_yaw_to_tf_matrix(np.array([1.0, 2.0, 3.0]), np.pi / 2)
    assert np.allclose(tf @ np.array([1.0, 0, 0, 1]), [1.0, 3.0, 3.0, 1.0])
    pts = np.array([[0.0, 0.0], [1.0, 0.0], [5.0, 5.0]])
    assert closest_point_within_threshold(pts, np.array([0.9, 0.0]), 0.5) == 1
    assert closest_point_within_threshold(pts, np.array([3.0, 3.0]), 0.5) == -1
    assert habitat_objectgoal_name(3) == "toilet"


def test_cycle_check_never_fires_like_the_reference():
    e = AcyclicEnforcer()
    e.add_state_action(np.zeros(2), np.ones(2), (0.3, 0.2))
    assert e.check_cyclic(np.zeros(2), np.ones(2), (0.3, 0.2)) is False  # identity comparison: acyclic_enforcer.py:8-17


def test_frontier_selector_rules():
    s = FrontierSelector()
    robot = np.zeros(2)
    pts = np.array([[1.0, 0.0], [2.0, 0.0], [3.0, 0.0]])
    best, val = s.choose(pts, [0.4, 0.3, 0.2], pts, robot)
    assert np.array_equal(best, pts[0]) and val == 0.4                       # nothing pursued yet: best value
    # the pursued frontier is still listed but another one is now better: stay while value + 0.01 > last value
    order = np.array([[2.0, 0.0], [1.0, 0.0], [3.0, 0.0]])
    best, val = s.choose(order, [0.5, 0.395, 0.2], order, robot)
    assert np.array_equal(best, [1.0, 0.0]) and val == 0.395
    # ... and leave it once it has dropped by more than 0.01 below the value it was last pursued at (0.395)
    best, val = s.choose(order, [0.5, 0.38, 0.2], order, robot)
    assert np.array_equal(best, [2.0, 0.0]) and val == 0.5
    # pursued frontier gone, one within 0.5 m of it takes its place (nearest, even if not the best)
    moved = np.array([[9.0, 9.0], [2.3, 0.0]])
    best, val = s.choose(moved, [0.9, 0.495], moved, robot)
    assert np.array_equal(best, [2.3, 0.0]) and val == 0.495
    # nothing near: best value again
    far = np.array([[9.0, 9.0], [7.0, 7.0]])
    best, val = s.choose(far, [0.2, 0.1], far, robot)
    assert np.array_equal(best, [9.0, 9.0])


def test_fixture_signatures_present():
    for name in POLICY_CASES:
        g = load(name)
        assert len(g["mode"]) == len(g["pose"]) == len(g["wall"])
        assert {"initialize", "explore"} <= set(g["mode"].tolist())


def _reference_maps():
    from oracle import ref_shim

    if not ref_shim.available():
        pytest.skip("/root/reference not present (GPU box): the fixtures are the pin there")
    vm, om, _, _ = ref_shim.reference_modules()
    return vm, om, ref_shim.reference_object_map(), ref_shim.reference_detections()


@pytest.mark.parametrize("name", POLICY_CASES)
def test_step_logic_over_reference_maps_matches_reference_policy(name):
    from vlfm_amd.policy_step import ITMPolicyV2Step

    vm, om, opm, det = _reference_maps()

    def make(vlm, **kw):
        obstacle = om.ObstacleMap(min_height=0.61, max_height=0.88, area_thresh=1.5, agent_radius=0.18,
                                  hole_area_thresh=100000)
        return ITMPolicyV2Step(itm=vlm.itm, coco_detector=vlm.coco, detector=vlm.gdino, sam=vlm.sam,
                               obstacle_map=obstacle, value_map=vm.ValueMap(value_channels=1, use_max_confidence=False),
                               object_map=opm.ObjectPointCloudMap(erosion_size=5), **kw)

    pol, g = replay_policy_episode(name, make, det.ObjectDetections, tol=0.0)
    obstacle, value, objects = pol.maps()
    assert np.array_equal(value._map, dense(g["conf_idx"], g["conf_val"], (1000, 1000), np.float32))
    assert sha(np.asarray(value._value_map, np.float64)) == str(g["value_sha"])
    assert np.array_equal(obstacle.explored_area.astype(bool), unpack_plane(g["explored"]))
    assert np.array_equal(obstacle._map.astype(bool), unpack_plane(g["obstacles"]))
    target = [k for k in objects.clouds]
    if target:
        assert sha(np.asarray(objects.clouds[target[0]], np.float64)) == str(g["cloud_sig"][1])


# ---------------------------------------------------------------------------------------------- branch coverage with stub maps
class _StubObstacle:
    def __init__(self, frontiers=None, raise_index=False):
        self.frontiers = np.zeros((0, 2)) if frontiers is None else frontiers
        self.raise_index, self.calls = raise_index, 0

    def reset(self):
        pass

    def update_map(self, *a):
        self.calls += 1
        if self.raise_index:
            raise IndexError("index 1000 is out of bounds for axis 0 with size 1000")  # obstacle_map.py:101

    def update_agent_traj(self, *a):
        pass


class _StubValue:
    def __init__(self):
        self.updates = []

    def reset(self):
        pass

    def update_map(self, values, *a):
        self.updates.append(np.asarray(values).copy())

    def update_agent_traj(self, *a):
        pass

    def sort_waypoints(self, wps, radius):
        vals = [0.1 * (i + 1) for i in range(len(wps))]
        order = np.argsort([-v for v in vals])
        return np.array([wps[i] for i in order]), [vals[i] for i in order]


class _StubObjects:
    def __init__(self, goal=None):
        self.goal, self.updates = goal, 0

    def reset(self):
        pass

    def has_object(self, name):
        return self.goal is not None

    def get_best_object(self, name, xy):
        return self.goal

    def update_map(self, *a):
        self.updates += 1

    def update_explored(self, *a):
        pass


class _Dets:
    def __init__(self, n=0):
        import torch

        self.boxes, self.logits, self.phrases = torch.rand(n, 4), torch.ones(n), ["chair"] * n

    num_detections = property(lambda self: len(self.logits))

    def filter_by_class(self, c):
        pass

    def filter_by_conf(self, t):
        pass


def _step_obj(obstacle, objects=None, n_det=0, **kw):
    from vlfm_amd.policy_step import ITMPolicyV2Step

    class Itm:
        def cosine(self, img, txt):
            return 0.3

    class Det:
        def predict(self, img, caption=""):
            return _Dets(n_det)

    class Sam:
        def segment_bbox(self, img, box):
            return np.ones(img.shape[:2], np.uint8)

    pol = ITMPolicyV2Step(camera_height=0.88, min_depth=0.5, max_depth=5.0, camera_fov=79.0, image_width=64, itm=Itm(),
                          coco_detector=Det(), detector=Det(), sam=Sam(), obstacle_map=obstacle, value_map=_StubValue(),
                          object_map=objects or _StubObjects(), **kw)
    pol.reset("chair")
    return pol


def test_step_branches_edge_of_map_no_frontiers_navigate_and_stop():
    rgb, depth = np.zeros((48, 64, 3), np.uint8), np.full((48, 64), 0.5, np.float32)
    # IndexError from the obstacle scatter = "Reached edge of map, stopping." (base_objectnav_policy.py:157-162)
    r = _step_obj(_StubObstacle(raise_index=True)).step(rgb, depth, 0.0, 0.0, 0.0)
    assert r.mode == "edge_of_map" and r.stop and r.goal is None
    # initialise for 12 steps, then explore with an empty frontier list -> STOP without called_stop (itm_policy.py:64-67)
    pol = _step_obj(_StubObstacle())
    modes = [pol.step(rgb, depth, 0.0, 0.0, 0.0) for _ in range(13)]
    assert [m.mode for m in modes[:12]] == ["initialize"] * 12 and modes[12].mode == "explore"
    assert modes[12].stop and not pol.called_stop and modes[12].goal is None
    # the reference's sentinel "no frontier" value is a single (0, 0) row as well
    pol = _step_obj(_StubObstacle(frontiers=np.zeros((1, 2))))
    assert [pol.step(rgb, depth, 0.0, 0.0, 0.0) for _ in range(13)][-1].stop
    # frontiers -> best value first; goal more than 0.1 m away from the previous one resets the controller
    fr = np.array([[1.0, 0.0], [0.0, 2.0], [3.0, 3.0]])
    pol = _step_obj(_StubObstacle(frontiers=fr))
    r = [pol.step(rgb, depth, 0.0, 0.0, 0.0) for _ in range(13)][-1]
    assert r.mode == "explore" and np.array_equal(r.goal, fr[2]) and r.pointnav_reset and not r.stop
    assert abs(r.rho - np.hypot(3, 3)) < 1e-12 and abs(r.theta - np.pi / 4) < 1e-12
    assert not pol.step(rgb, depth, 0.0, 0.0, 0.0).pointnav_reset                     # same goal again
    # an object goal: navigate; inside pointnav_stop_radius -> STOP and called_stop
    pol = _step_obj(_StubObstacle(frontiers=fr), objects=_StubObjects(goal=np.array([0.5, 0.0])), n_det=1)
    r = [pol.step(rgb, depth, 0.0, 0.0, 0.0) for _ in range(13)][-1]
    assert r.mode == "navigate" and r.stop and pol.called_stop and pol.maps()[2].updates == 13
    assert int(pol.object_masks.sum()) == 48 * 64


def test_step_all_ones_depth_needs_infer_depth_like_the_reference():
    rgb, ones = np.zeros((48, 64, 3), np.uint8), np.ones((48, 64), np.float32)
    with pytest.raises(NotImplementedError):           # BaseObjectNavPolicy._infer_depth base_objectnav_policy.py:358-365
        _step_obj(_StubObstacle(), n_det=1).step(rgb, ones, 0.0, 0.0, 0.0)
    pol = _step_obj(_StubObstacle(), n_det=1, infer_depth=lambda rgb, lo, hi: np.full((48, 64), 0.4, np.float32))
    assert pol.step(rgb, ones, 0.0, 0.0, 0.0).mode == "initialize"
    with pytest.raises(NotImplementedError):
        _step_obj(_StubObstacle(), use_vqa=True)


def test_multi_prompt_value_channels_and_pipe_substitution():
    from vlfm_amd.policy_step import ITMPolicyV2Step

    seen = []

    class Itm:
        def cosine(self, img, txt):
            seen.append(txt)
            return 0.1 * len(seen)

    vm = _StubValue()
    pol = ITMPolicyV2Step(camera_height=0.88, min_depth=0.5, max_depth=5.0, camera_fov=79.0, image_width=64,
                          text_prompt="Seems like there is a target_object ahead.|There is a lot of area to explore ahead.",
                          itm=Itm(), coco_detector=None or type("D", (), {"predict": lambda s, i, caption="": _Dets(0)})(),
                          detector=type("D", (), {"predict": lambda s, i, caption="": _Dets(0)})(),
                          sam=object(), obstacle_map=_StubObstacle(), value_map=vm, object_map=_StubObjects())
    pol.reset("table|desk")
    pol.step(np.zeros((48, 64, 3), np.uint8), np.full((48, 64), 0.5, np.float32), 0.0, 0.0, 0.0)
    assert seen == ["Seems like there is a table/desk ahead.", "There is a lot of area to explore ahead."]
    assert vm.updates[0].shape == (2,) and np.allclose(vm.updates[0], [0.1, 0.2])


def test_v3_scores_by_exploration_channel_below_the_threshold():
    """ITMPolicyV3._reduce_values (itm_policy.py:296-318), checked against the reference's own method when available."""
    from vlfm_amd.policy_step import ITMPolicyV3Step

    class TwoChannel(_StubValue):
        def sort_waypoints(self, wps, radius, reduce_fn=None):
            vals = reduce_fn([(0.10, 0.9), (0.30, 0.2), (0.20, 0.5)])
            order = np.argsort([-v for v in vals])
            return np.array([wps[i] for i in order]), [vals[i] for i in order]

    fr = np.array([[1.0, 0.0], [0.0, 2.0], [3.0, 3.0]])
    kw = dict(camera_height=0.88, min_depth=0.5, max_depth=5.0, camera_fov=79.0, image_width=64,
              text_prompt="a|b", itm=type("I", (), {"cosine": lambda s, i, t: 0.2})(),
              coco_detector=type("D", (), {"predict": lambda s, i, caption="": _Dets(0)})(),
              detector=type("D", (), {"predict": lambda s, i, caption="": _Dets(0)})(), sam=object(),
              obstacle_map=_StubObstacle(frontiers=fr), object_map=_StubObjects())
    rgb, depth = np.zeros((48, 64, 3), np.uint8), np.full((48, 64), 0.5, np.float32)
    for thresh, want in ((0.25, fr[1]), (0.35, fr[0])):      # target channel decides / exploration channel decides
        pol = ITMPolicyV3Step(thresh, value_map=TwoChannel(), **kw)
        pol.reset("chair")
        goal = [pol.step(rgb, depth, 0.0, 0.0, 0.0) for _ in range(13)][-1].goal
        assert np.array_equal(goal, want), (thresh, goal)
    from oracle import ref_shim

    if ref_shim.available():
        itm_mod, _ = ref_shim.reference_policy()
        ref = itm_mod.ITMPolicyV3._reduce_values
        fake = type("P", (), {"_exploration_thresh": 0.25})()
        vals = [(0.10, 0.9), (0.30, 0.2), (0.20, 0.5)]
        assert ref(fake, vals) == ITMPolicyV3Step._reduce_values(fake, vals)
        fake._exploration_thresh = 0.35
        assert ref(fake, vals) == ITMPolicyV3Step._reduce_values(fake, vals)


def test_step_cameras_follows_the_robot_deployment_call_order():
    """reality_policies.py:103-141 + itm_policy.py:191-211: obstacle updates (explore=False for all but the last entry, which
    only reveals), every cosine before the first value-map update, one object-map update per camera."""
    from vlfm_amd.policy_step import ITMPolicyV2Step

    log = []

    class Obstacle(_StubObstacle):
        def update_map(self, depth, tf, lo, hi, fx, fy, fov, explore=True, update_obstacles=True):
            log.append(("obstacle", depth is None, explore, update_obstacles))

    class Value(_StubValue):
        def update_map(self, values, depth, tf, lo, hi, fov):
            log.append(("value", float(values[0]), hi))

    class Itm:
        n = 0

        def cosine(self, img, txt):
            Itm.n += 1
            log.append(("cosine", int(img[0, 0, 0])))
            return 0.1 * Itm.n

    class Objects(_StubObjects):
        def update_explored(self, tf, max_depth, fov):
            log.append(("object", max_depth, round(float(fov), 6)))

    det = type("D", (), {"predict": lambda s, i, caption="": _Dets(0)})()
    pol = ITMPolicyV2Step(camera_height=0.88, min_depth=0.5, max_depth=5.0, camera_fov=79.0, image_width=64, itm=Itm(),
                          coco_detector=det, detector=det, sam=object(), obstacle_map=Obstacle(), value_map=Value(),
                          object_map=Objects())
    pol.reset("chair")
    tf = np.eye(4)
    cams = [np.full((4, 6, 3), k, np.uint8) for k in range(2)]
    d = np.full((4, 6), 0.5, np.float32)
    r = pol.step_cameras(obstacle_map_depths=[(d, tf, 0.3, 3.0, 5.0, 5.0, 1.0), (d, tf, 0.3, 3.0, 5.0, 5.0, 1.0),
                                              (None, tf, 0.3, 3.0, 5.0, 5.0, 1.2)],
                         value_map_rgbd=[(cams[0], d, tf, 0.3, 3.0, 1.0), (cams[1], d, tf, 0.3, 2.5, 1.0)],
                         object_map_rgbd=[(cams[0], d, tf, 0.3, 3.0, 5.0, 5.0), (cams[1], d, tf, 0.3, 2.5, 4.0, 4.0)],
                         robot_xy=np.zeros(2), robot_heading=0.0, nav_depth=d)
    assert r.mode == "initialize"
    assert log == [("obstacle", False, False, True), ("obstacle", False, False, True), ("obstacle", True, True, False),
                   ("cosine", 0), ("cosine", 1), ("value", 0.1, 3.0), ("value", 0.2, 2.5),
                   ("object", 3.0, round(2 * np.arctan(3 / 5.0), 6)), ("object", 2.5, round(2 * np.arctan(3 / 4.0), 6))]

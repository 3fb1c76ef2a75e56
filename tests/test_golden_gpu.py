"""-m gpu: the HIP path (through the C ABI) against the golden fixtures that the REFERENCE'S OWN SOURCE produced
(tests/golden/make_golden.py).  BASELINE.json's bar is 1e-4 on the float maps; since round 3 the device keeps the value map
in f64 exactly as the reference's array ends up (value_map.py:423), so the bar HERE is equality: confidence map (f32) and
value map (f64 / f32 by the reference's own dtype rule) bit for bit, sort_waypoints values and permutation exact, obstacle /
navigable / explored planes, frontier pixels and world coordinates bit-exact.  Nothing here reads /root/reference: the
fixtures travel with the repo."""
import numpy as np
import pytest

from golden_util import OM_CASES, VM_CASES, dense, frames, load, split_frontiers, unpack_plane

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("name", VM_CASES)
def test_value_map_matches_reference_fixture(gpu_device, name):
    from vlfm_amd.mapping import ValueMap

    g = load(name)
    C, H, W = int(g["channels"]), int(g["height"]), int(g["width"])
    vm = ValueMap(C, use_max_confidence=bool(g["use_max_confidence"]), fusion_type=str(g["fusion_type"]),
                  device=gpu_device)
    for depth, tf, values in frames(g, C, height=H, width=W):
        vm.update_map(values, depth, tf, float(g["min_depth"]), float(g["max_depth"]), float(g["fov"]))
    conf = dense(g["conf_idx"], g["conf_val"], (1000, 1000), np.float32)
    value = dense(g["value_idx"], g["value_val"], (1000, 1000, C), np.float64)
    got_c, got_v = vm._map, vm._value_map
    assert got_c.dtype == np.float32 and str(got_v.dtype) == str(g["value_dtype"])  # the reference's f32 -> f64 drift
    assert np.array_equal(got_c, conf), np.abs(got_c - conf).max()
    assert np.array_equal(got_v.astype(np.float64), value), np.abs(got_v - value).max()
    red = None if C == 1 else (lambda vs: [max(v) for v in vs])
    s_wp, s_val = vm.sort_waypoints(g["waypoints"], 0.5, reduce_fn=red)
    assert np.array_equal(np.asarray(s_wp), g["sorted_waypoints"])  # bit-exact frontier order
    assert np.array_equal(np.asarray(s_val, np.float64), g["sorted_values"])


@pytest.mark.parametrize("name", OM_CASES)
def test_obstacle_map_matches_reference_fixture(gpu_device, name):
    from vlfm_amd.mapping import ObstacleMap

    g = load(name)
    om = ObstacleMap(min_height=float(g["min_height"]), max_height=float(g["max_height"]),
                     agent_radius=float(g["agent_radius"]), area_thresh=float(g["area_thresh"]),
                     hole_area_thresh=int(g["hole_area_thresh"]), device=gpu_device)
    want_px, want_xy = split_frontiers(g)
    for k, (depth, tf, _) in enumerate(frames(g, holes=bool(g["holes"]))):
        om.update_map(depth, tf, float(g["min_depth"]), float(g["max_depth"]), float(g["fx"]), float(g["fy"]),
                      float(g["fov"]))
        assert np.array_equal(np.asarray(om._frontiers_px, np.float64).reshape(-1, 2), want_px[k]), f"step {k}"
        assert np.array_equal(np.asarray(om.frontiers, np.float64).reshape(-1, 2), want_xy[k]), f"step {k}"
    assert np.array_equal(om._map.astype(bool), unpack_plane(g["obstacle_bits"]))
    assert np.array_equal(om._navigable_map.astype(bool), unpack_plane(g["navigable_bits"]))
    assert np.array_equal(om.explored_area.astype(bool), unpack_plane(g["explored_bits"]))


def test_sync_explored_matches_reference_fixture(gpu_device):
    from vlfm_amd.mapping import ObstacleMap, ValueMap

    g = load("vm_sync_explored")
    om = ObstacleMap(min_height=float(g["min_height"]), max_height=float(g["max_height"]),
                     agent_radius=float(g["agent_radius"]), area_thresh=float(g["area_thresh"]), device=gpu_device)
    vm = ValueMap(1, use_max_confidence=False, obstacle_map=om, device=gpu_device)
    for depth, tf, values in frames(g):
        om.update_map(depth, tf, float(g["min_depth"]), float(g["max_depth"]), float(g["fx"]), float(g["fy"]),
                      float(g["fov"]))
        vm.update_map(values, depth, tf, float(g["min_depth"]), float(g["max_depth"]), float(g["fov"]))
    conf = dense(g["conf_idx"], g["conf_val"], (1000, 1000), np.float32)
    value = dense(g["value_idx"], g["value_val"], (1000, 1000, 1), np.float64)
    assert np.array_equal(om.explored_area.astype(bool), unpack_plane(g["explored_bits"]))
    assert np.array_equal(vm._map, conf) and np.array_equal(vm._value_map, value)


def test_multicamera_obstacle_map_matches_reference_fixture(gpu_device):
    """The robot deployment's call pattern: obstacles from two body cameras (explore=False), then a reveal from the robot
    pose without a depth image (update_obstacles=False)."""
    from golden_util import replay_multicam
    from vlfm_amd.mapping import ObstacleMap

    replay_multicam(lambda **kw: ObstacleMap(device=gpu_device, **kw))


def test_depth_islands_inside_small_holes_match_reference_fixture(gpu_device):
    """fill_small_holes draws each small zero-region contour FILLED, so valid texels it encloses are dropped too
    (img_utils.py:385-388): the speculative single depth pass has to take their obstacle bits back."""
    from golden_util import replay_islands
    from vlfm_amd.mapping import ObstacleMap

    replay_islands(lambda **kw: ObstacleMap(device=gpu_device, **kw))


def test_500_step_episode_matches_reference_fixture(gpu_device):
    """BASELINE's own episode length (500 steps), EXACT: frontier pixels at every step (13-22 simultaneous frontiers), the
    sort_waypoints values and permutation at every step (np.argsort over f64 medians of the f64-promoted value map,
    value_map.py:183), planes at steps 100 / 250 / 500, and the SHA-256 of the f32 confidence map and of the f64 value map
    at those steps equal to the digests of the reference's arrays."""
    from golden_util import replay_episode500
    from vlfm_amd.mapping import ObstacleMap, ValueMap

    replay_episode500(lambda **kw: ObstacleMap(device=gpu_device, **kw),
                      lambda c, **kw: ValueMap(c, device=gpu_device, **kw))


def test_two_camera_value_map_matches_reference_fixture(gpu_device):
    """Two cameras with different (fov, max_depth) feeding one value map: one cone template per optics, like the
    reference's per-(fov, max_depth) confidence-mask cache."""
    from golden_util import replay_two_cameras
    from vlfm_amd.mapping import ValueMap

    replay_two_cameras(lambda c, **kw: ValueMap(c, device=gpu_device, **kw))

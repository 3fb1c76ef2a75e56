"""CPU: pins the oracle to the REFERENCE'S OWN SOURCE.

tests/golden/*.npz were produced by running /root/reference/vlfm/mapping/{value_map,obstacle_map}.py and
vlfm/utils/{geometry_utils,img_utils}.py themselves (oracle/ref_shim.py supplies stand-ins for the absent cv2 and
frontier_exploration packages).  Here the oracle restatement must reproduce every fixture BIT-EXACTLY (same cv2
stand-in underneath, so any difference is a restatement error in the in-tree arithmetic), and -- when /root/reference is
present -- the committed fixtures must equal a fresh run of the reference."""
import os
import subprocess
import sys

import numpy as np
import pytest

from golden_util import OM_CASES, VM_CASES, dense, frames, load, split_frontiers, unpack_plane

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", VM_CASES)
def test_oracle_value_map_reproduces_reference_fixture(name):
    from oracle.ref_value_map import RefValueMap

    g = load(name)
    C, H, W = int(g["channels"]), int(g["height"]), int(g["width"])
    vm = RefValueMap(C, use_max_confidence=bool(g["use_max_confidence"]), fusion_type=str(g["fusion_type"]))
    for depth, tf, values in frames(g, C, height=H, width=W):
        vm.update_map(values, depth.copy(), tf, float(g["min_depth"]), float(g["max_depth"]), float(g["fov"]))
    assert str(vm._value_map.dtype) == str(g["value_dtype"])  # the reference's f32 -> f64 drift included
    assert np.array_equal(vm._map, dense(g["conf_idx"], g["conf_val"], (1000, 1000), np.float32))
    assert np.array_equal(vm._value_map, dense(g["value_idx"], g["value_val"], (1000, 1000, C), np.float64))
    red = None if C == 1 else (lambda vs: [max(v) for v in vs])
    s_wp, s_val = vm.sort_waypoints(g["waypoints"], 0.5, reduce_fn=red)
    assert np.array_equal(np.asarray(s_wp), g["sorted_waypoints"])
    assert np.array_equal(np.asarray(s_val, np.float64), g["sorted_values"])


@pytest.mark.parametrize("name", OM_CASES)
def test_oracle_obstacle_map_reproduces_reference_fixture(name):
    from oracle.ref_obstacle_map import RefObstacleMap

    g = load(name)
    om = RefObstacleMap(min_height=float(g["min_height"]), max_height=float(g["max_height"]),
                        agent_radius=float(g["agent_radius"]), area_thresh=float(g["area_thresh"]),
                        hole_area_thresh=int(g["hole_area_thresh"]))
    want_px, want_xy = split_frontiers(g)
    for k, (depth, tf, _) in enumerate(frames(g, holes=bool(g["holes"]))):
        om.update_map(depth.copy(), tf, float(g["min_depth"]), float(g["max_depth"]), float(g["fx"]), float(g["fy"]),
                      float(g["fov"]))
        assert np.array_equal(np.asarray(om._frontiers_px, np.float64).reshape(-1, 2), want_px[k]), f"step {k}"
        assert np.array_equal(np.asarray(om.frontiers, np.float64).reshape(-1, 2), want_xy[k]), f"step {k}"
    assert np.array_equal(om._map.astype(bool), unpack_plane(g["obstacle_bits"]))
    assert np.array_equal(np.asarray(om._navigable_map).astype(bool), unpack_plane(g["navigable_bits"]))
    assert np.array_equal(om.explored_area.astype(bool), unpack_plane(g["explored_bits"]))


def test_oracle_sync_explored_reproduces_reference_fixture():
    from oracle.ref_obstacle_map import RefObstacleMap
    from oracle.ref_value_map import RefValueMap

    g = load("vm_sync_explored")
    om = RefObstacleMap(min_height=float(g["min_height"]), max_height=float(g["max_height"]),
                        agent_radius=float(g["agent_radius"]), area_thresh=float(g["area_thresh"]))
    vm = RefValueMap(1, use_max_confidence=False, obstacle_map=om)
    for depth, tf, values in frames(g):
        om.update_map(depth.copy(), tf, float(g["min_depth"]), float(g["max_depth"]), float(g["fx"]), float(g["fy"]),
                      float(g["fov"]))
        vm.update_map(values, depth.copy(), tf, float(g["min_depth"]), float(g["max_depth"]), float(g["fov"]))
    assert np.array_equal(vm._map, dense(g["conf_idx"], g["conf_val"], (1000, 1000), np.float32))
    assert np.array_equal(vm._value_map, dense(g["value_idx"], g["value_val"], (1000, 1000, 1), np.float64))
    assert np.array_equal(om.explored_area.astype(bool), unpack_plane(g["explored_bits"]))


def test_helpers_match_reference_fixture():
    """oracle/ref_geometry.py + the product's host-side BaseMap against geometry_utils / img_utils / base_map."""
    from oracle import ref_geometry as rg
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.synthetic import pose_to_tf

    g = load("helpers")
    for p, y, tf, ey in zip(g["xyz"], g["yaws"], g["tfs"], g["extract_yaw"]):
        assert np.array_equal(pose_to_tf(p[0], p[1], y, p[2]), tf)  # xyz_yaw_to_tf_matrix
        assert rg.yaw_of(tf) == ey                                    # extract_yaw
    cloud = rg.unproject(g["depth"], g["mask"], 7.5, 7.25)
    assert np.array_equal(cloud, g["cloud"])
    assert np.array_equal(rg.apply_tf(g["tfs"][0], cloud), g["moved"])
    bm = RefValueMap(1, size=200)
    assert np.array_equal(bm._xy_to_px(g["pts"]), g["px"])
    assert np.array_equal(bm._px_to_xy(g["px"]), g["back"])
    med = np.array([rg.disc_reduce(g["field"], tuple(c), 10) for c in g["cells"]], np.float64)
    assert np.array_equal(med, g["medians"])
    rot = np.stack([rg.rotate_about_centre(g["tile"], a) for a in (0.0, 0.3, -1.1, np.pi / 2)])
    assert np.array_equal(rot, g["rotated"])
    assert np.array_equal(rg.paste_centred(np.zeros((30, 30), np.float32), g["tile"], 4, 27), g["placed"])


def test_product_basemap_host_logic_matches_reference_fixture():
    """vlfm_amd.mapping.BaseMap._xy_to_px/_px_to_xy are host code of the product (no GPU needed)."""
    from vlfm_amd.mapping.base_map import BaseMap

    g = load("helpers")
    bm = BaseMap(200)
    assert np.array_equal(bm._xy_to_px(g["pts"]), g["px"])
    assert np.array_equal(bm._px_to_xy(g["px"]), g["back"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/vlfm"), reason="reference tree not present (GPU box)")
def test_committed_fixtures_equal_a_fresh_run_of_the_reference():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--check"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_oracle_matches_multicamera_fixture():
    """explore=False per body camera, then a reveal without depth (reality_policies.py:113-138)."""
    from golden_util import replay_multicam
    from oracle.ref_obstacle_map import RefObstacleMap

    replay_multicam(lambda **kw: RefObstacleMap(**kw))


def test_oracle_matches_two_camera_value_map_fixture():
    from golden_util import replay_two_cameras
    from oracle.ref_value_map import RefValueMap

    replay_two_cameras(lambda c, **kw: RefValueMap(c, **kw))


def test_oracle_matches_depth_island_fixture():
    from golden_util import replay_islands
    from oracle.ref_obstacle_map import RefObstacleMap

    replay_islands(lambda **kw: RefObstacleMap(**kw))


def test_oracle_reproduces_the_500_step_episode_fixture():
    """Full episode length of the reference (pointnav_depth_hm3d.yaml:14): 13-22 simultaneous frontiers, pillars and
    non-convex blocks in the cone; every step bit-exact, maps at steps 100 / 250 / 500 by digest."""
    from golden_util import replay_episode500
    from oracle.ref_obstacle_map import RefObstacleMap
    from oracle.ref_value_map import RefValueMap

    replay_episode500(lambda **kw: RefObstacleMap(**kw), lambda c, **kw: RefValueMap(c, **kw))

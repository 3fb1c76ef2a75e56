"""Host half of the C ABI (scalar f64 prologues, CPU-callable) against the oracle's restatement."""
import ctypes

import numpy as np
import pytest

from oracle import cv
from oracle.ref_geometry import pose_to_tf, yaw_of
from oracle.ref_value_map import RefValueMap
from vlfm_amd import _lib
from vlfm_amd.mapping.value_map import pose_params

FOV = np.deg2rad(79)


def test_pose_params_match_oracle_rotation_and_cell():
    rng = np.random.default_rng(0)
    tfs = np.stack([pose_to_tf((rng.uniform(-20, 20), rng.uniform(-20, 20), 0.88), rng.uniform(-np.pi, np.pi))
                    for _ in range(64)] + [pose_to_tf((0.049, -0.049, 0.88), np.pi / 6 * k) for k in range(12)])
    out = pose_params(tfs, None, 1000, 20, 201)
    for k, tf in enumerate(tfs):
        M = cv.getRotationMatrix2D((100, 100), np.degrees(-yaw_of(tf)), 1.0).reshape(-1).copy()
        D = M[0] * M[4] - M[1] * M[3]
        D = 1.0 / D if D != 0 else 0
        A11, A22 = M[4] * D, M[0] * D
        M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22
        b1 = -M[0] * M[2] - M[1] * M[5]; b2 = -M[3] * M[2] - M[4] * M[5]
        M[2] = b1; M[5] = b2
        assert out[k]['inv_affine'].tolist() == M.tolist()      # bit-exact f64
        px = int(tf[0, 3] * 20) + 500
        py = int(-tf[1, 3] * 20) + 500
        assert (out[k]['row0'], out[k]['col0'], out[k]['env']) == (px - 100, py - 100, k)


def test_pose_params_outside_map_is_the_reference_assertion():
    with pytest.raises(AssertionError, match="Pixel location is outside the image."):
        pose_params(pose_to_tf((25.05, 0, 0.88), 0.0)[None], None, 1000, 20, 201)
    with pytest.raises(AssertionError):
        pose_params(pose_to_tf((0, 25.05, 0.88), 0.0)[None], None, 1000, 20, 201)
    pose_params(pose_to_tf((24.99, -24.99, 0.88), 0.0)[None], None, 1000, 20, 201)


@pytest.mark.parametrize("fov_deg,max_depth", [(79.0, 5.0), (60.0, 3.5), (90.0, 10.0)])
def test_cone_template_host_matches_oracle(fov_deg, max_depth):
    fov = np.deg2rad(fov_deg)
    T = 2 * int(max_depth * 20) + 1
    conf = np.zeros(T * T, np.float32)
    poly = np.zeros(1024, np.int64)
    n = ctypes.c_int(0)
    rc = _lib.lib().vlfm_cone_template_host(fov, max_depth, 20, 0.25, conf.ctypes.data, conf.size, poly.ctypes.data,
                                            512, ctypes.byref(n))
    assert rc == T
    want_poly = cv.ellipse_polygon((T // 2, T // 2), (T // 2, T // 2), 0, -np.rad2deg(fov) / 2 + 90,
                                   np.rad2deg(fov) / 2 + 90)
    assert np.array_equal(poly[: 2 * n.value].reshape(-1, 2), want_poly)
    RefValueMap._confidence_masks.pop((fov, max_depth), None)
    ref = RefValueMap(1)._get_confidence_mask(fov, max_depth)
    inside = ref > 0
    got = conf.reshape(T, T)
    # NumPy's scalar trig may differ from libm by an ulp of f64 before the f32 store: allow one f32 ulp
    assert np.abs(got[inside] - ref[inside]).max() <= 2 ** -23
    assert (got[inside] == ref[inside].astype(np.float32)).mean() > 0.999


def test_tan_table_and_disc_rows():
    tab = np.zeros(640)
    _lib.lib().vlfm_tan_table_host(FOV, 640, tab.ctypes.data)
    want = np.tan(np.linspace(-FOV / 2, FOV / 2, 640))
    assert np.abs(tab - want).max() <= 4 * np.finfo(np.float64).eps
    assert tab[-1] == np.tan(FOV / 2) or abs(tab[-1] - np.tan(FOV / 2)) < 1e-15
    for r in (1, 2, 5, 10, 17):
        hw = np.zeros(2 * r + 1, np.int32)
        _lib.lib().vlfm_disc_rows_host(r, hw.ctypes.data)
        disc = np.zeros((2 * r + 1, 2 * r + 1), np.uint8)
        cv.circle(disc, (r, r), r, 255, -1)
        for i in range(2 * r + 1):
            xs = np.where(disc[i])[0]
            assert xs[0] == r - hw[i] and xs[-1] == r + hw[i] and len(xs) == 2 * hw[i] + 1


def test_harness_decide_and_navigate_are_the_scalar_rules_vectorised():
    """BatchedEpisodes._decide / _navigate (host side of the full step) against the scalar rules of policy_step.py: the three
    modes of base_objectnav_policy.py:126-135, frontier stickiness (itm_policy.py:76-152), goal hand-over and stop rule
    (:243-283), and the controller state of environments that do not consult the controller."""
    import numpy as np
    import torch

    from vlfm_amd.harness import BatchedEpisodes
    from vlfm_amd.policy_step import ACTION_STOP, ACTION_TURN_LEFT, FrontierSelector, rho_theta

    class H:  # just the attributes the two methods touch
        pass

    class FakeObjectMap:
        def __init__(self, goal):
            self.goal, self.asked = goal, 0

        def has_object(self, name):
            return self.goal is not None

        def get_best_object(self, name, xy):
            self.asked += 1
            return np.asarray(self.goal)

    h = H()
    h.E, h.targets, h.device, h.stop_radius = 4, ["chair", "bed", "tv", "couch"], torch.device("cpu"), 0.9
    h.sightings = None                                     # no episode script: every environment is on the harness clock
    h._episode_steps = lambda t_ep: BatchedEpisodes._episode_steps(h, t_ep)
    h.selectors = [FrontierSelector() for _ in range(4)]
    h.object_maps = [FakeObjectMap(None), FakeObjectMap(None), FakeObjectMap(None), FakeObjectMap([4.5, 4.0, 0.3])]
    wps = np.array([[1.0, 0.0], [2.0, 0.0], [5.0, 5.0], [6.0, 6.0], [7.0, 7.0]])
    env_of = np.array([0, 0, 2, 2, 2])
    poses = np.array([[0.0, 0.0, 0.3], [1.0, 1.0, -1.0], [4.0, 4.0, 2.0], [4.0, 4.0, 0.1]])
    vals = np.array([0.2, 0.3, 0.1, 0.5, 0.4])
    # during the 12 initialisation steps nobody gets a goal, but the object map is consulted like in the reference (:123)
    modes, goals, halt = BatchedEpisodes._decide(h, wps, env_of, vals, poses, 5)
    assert modes == ["initialize"] * 4 and np.isnan(goals).all() and not halt.any() and h.object_maps[3].asked == 1
    assert np.array_equal(h.selectors[0].last_frontier, np.zeros(2))            # selectors untouched while initialising
    modes, goals, halt = BatchedEpisodes._decide(h, wps, env_of, vals, poses, 12)
    assert modes == ["explore", "explore", "explore", "navigate"]
    assert np.array_equal(goals[0], [2.0, 0.0]) and np.isnan(goals[1]).all() and np.array_equal(goals[2], [6.0, 6.0])
    assert np.array_equal(goals[3], [4.5, 4.0]) and halt.tolist() == [False, True, False, False]   # env 1: no frontier -> stop
    # next step: env 0's pursued frontier dropped by less than 0.01 -> stays; env 2's is gone -> nearest within 0.5 m
    modes2, goals2, halt2 = BatchedEpisodes._decide(h, np.array([[1.0, 0.0], [2.0, 0.0], [6.2, 6.0], [9.0, 9.0]]),
                                                    np.array([0, 0, 2, 2]), np.array([0.9, 0.295, 0.495, 0.8]), poses, 13)
    assert np.array_equal(goals2[0], [2.0, 0.0]) and np.array_equal(goals2[2], [6.2, 6.0])

    captured = {}

    class Ctrl:
        discrete = True
        pointnav_test_recurrent_hidden_states = torch.arange(4.0).reshape(4, 1, 1).repeat(1, 2, 3)
        pointnav_prev_actions = torch.arange(4).reshape(4, 1)

        def reset(self, ids):
            captured["reset"] = list(np.asarray(ids))

        def act_on_depth(self, depth, rt, masks):
            captured["rt"], captured["masks"] = rt.numpy(), masks.numpy()
            self.pointnav_test_recurrent_hidden_states = self.pointnav_test_recurrent_hidden_states + 100.0
            self.pointnav_prev_actions = torch.full((4, 1), 1)
            return torch.ones(4, 1, dtype=torch.long)

    h.pointnav, h.prev_goals = Ctrl(), np.zeros((4, 2))
    h.prev_goals[0] = goals2[0]  # env 0 keeps its goal -> no reset; env 1 has no goal; env 2 moved; env 3 navigates, within 0.9 m
    acts = BatchedEpisodes._navigate(h, torch.zeros(4, 4, 4), modes2, goals2, halt2, poses)
    for e in (0, 2):
        rho, theta = rho_theta(poses[e, :2], poses[e, 2], goals2[e])
        assert abs(captured["rt"][e, 0] - rho) < 1e-6 and abs(captured["rt"][e, 1] - theta) < 1e-6
    assert captured["reset"] == [2, 3] and captured["masks"].tolist() == [True, True, False, False]
    assert acts.tolist() == [1, ACTION_STOP, 1, ACTION_STOP]     # no frontier -> STOP; object goal 0.5 m away -> STOP
    assert h.last_stops.tolist() == [False, True, False, True] and np.isnan(h.last_rho_theta[1]).all()
    st = h.pointnav.pointnav_test_recurrent_hidden_states[:, 0, 0].tolist()
    assert st == [100.0, 1.0, 102.0, 3.0]                        # stopped environments keep their controller state
    assert h.pointnav.pointnav_prev_actions.reshape(-1).tolist() == [1, 1, 1, 3]
    modes0 = ["initialize"] * 4
    acts0 = BatchedEpisodes._navigate(h, torch.zeros(4, 4, 4), modes0, np.full((4, 2), np.nan), np.zeros(4, bool), poses)
    assert acts0.tolist() == [ACTION_TURN_LEFT] * 4


def test_episode_log_matches_the_reference_log_saver_fixture(tmp_path, monkeypatch, capsys):
    """vlfm/utils/log_saver.py:9-44: same file name, same bytes, never overwrites, is_evaluated semantics --
    against what the reference's own module wrote (tests/golden/episode_log.npz)."""
    import os
    import sys
    import time

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden as mg
    from golden_util import load
    from vlfm_amd.utils.log_saver import is_evaluated, log_episode

    g = load("episode_log")
    monkeypatch.setenv("ZSOS_LOG_DIR", str(tmp_path / "logs"))
    assert is_evaluated(17, "scene_a") == bool(g["evaluated_before"]) is False
    log_episode(17, "scene_a", mg.EPISODE_LOG_SAMPLE)
    log_episode(17, "scene_a", {"failure_cause": "overwritten?"})
    assert is_evaluated(17, "scene_a") == bool(g["evaluated_after"]) is True
    names = sorted(os.listdir(tmp_path / "logs"))
    assert names == [str(n) for n in g["file_names"]]
    assert open(tmp_path / "logs" / names[0]).read() == str(g["text"])
    assert "Logging episode 0017" in capsys.readouterr().out
    # empty files older than five minutes are swept, fresh ones are kept and count as "being evaluated"
    stale, fresh = tmp_path / "logs" / "3_scene_b.json", tmp_path / "logs" / "4_scene_b.json"
    stale.write_text("")
    fresh.write_text("")
    os.utime(stale, (time.time() - 400, time.time() - 400))
    assert is_evaluated(4, "scene_b") and not stale.exists() and not is_evaluated(3, "scene_b")


def test_obstacle_map_dirty_window_bookkeeping():
    """ObstacleMapBatch's dirty windows (csrc/obstacle_map.hip: navigable_kernel / frontier_prepare_kernel): a fresh or reset
    slot hands over the whole map; a frame's reach window bounds every cell its scatter can touch (checked against the
    reference's own unprojection, geometry_utils.py:205-236 + base_map.py:44-46); a window that leaves the map, or a
    non-rigid transform, widens to the whole map (NumPy's negative-index wrap lands on the far side); windows accumulate
    over explore=False calls and are consumed by the call that uses them; the explored-mask window is the mirror of the
    revealed area's bounding box BEFORE the call, the refresh window contains it AFTER the call grown by 3; the launch sizes
    cover the largest window of the batch."""
    from oracle.ref_geometry import apply_tf as transform_points, unproject as get_point_cloud
    from vlfm_amd.mapping.obstacle_map import ObstacleMapBatch
    from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, camera_intrinsics, pose_to_tf

    S, E, R = 1000, 4, 100
    ob = ObstacleMapBatch.__new__(ObstacleMapBatch)   # host bookkeeping only: no device
    ob.size, ob.pixels_per_meter, ob.kernel_size, ob.n_envs = S, 20, 7, E
    full, empty = np.array([0, S - 1, 0, S - 1], np.int32), np.array([0, -1, 0, -1], np.int32)
    ob._dirty_obst = np.tile(full, (E, 1))
    ob._dirty_nav = ob._dirty_obst.copy()
    ob._bbox_host = np.tile(empty, (E, 1))
    env = np.arange(E)
    agent = np.array([[500, 500], [420, 610], [900, 510], [500, 500]])          # (x = col, y = row)
    w, nb_nav, nb_prep = ob._take_windows(env, True, True, agent, R)
    assert (w[:, 0:4] == full).all() and (w[:, 8:12] == full).all()              # first step = the full-map pass
    assert (w[:, 5] < w[:, 4]).all()                                             # nothing revealed yet: nothing to mask
    assert nb_nav == nb_prep == -(-S * 32 // 256)
    assert (ob._dirty_obst[:, 1] < ob._dirty_obst[:, 0]).all() and (ob._dirty_nav[:, 1] < ob._dirty_nav[:, 0]).all()
    assert (ob._bbox_host[0] == (398, 602, 398, 602)).all() and (ob._bbox_host[2] == (408, 612, 798, 999)).all()
    # nothing ingested: only the explore half has work -- mask inside the old box, refresh inside the new box + 3
    w, nb_nav, nb_prep = ob._take_windows(env, True, True, agent + 5, R)
    assert (w[:, 1] < w[:, 0]).all()
    assert (w[0, 4:8] == (398, 602, 398, 602)).all() and (w[0, 8:12] == (395, 610, 395, 610)).all()
    area = lambda q: int(((q[:, 1] - q[:, 0] + 1) * ((q[:, 3] >> 5) - (q[:, 2] >> 5) + 1)).max())   # rows x 32-cell words
    assert nb_nav == -(-area(w[:, 4:8]) // 256) == 7 and nb_prep == -(-area(w[:, 8:12]) // 256)

    H, W = 480, 640
    fx, fy, _ = camera_intrinsics(W)
    reach = MAX_DEPTH * float(np.sqrt(1.0 + (W / 2 / fx) ** 2 + (H / 2 / fy) ** 2)) * 20
    rng = np.random.default_rng(5)
    tf = np.stack([pose_to_tf(3.0, -2.0, 0.7), pose_to_tf(-6.5, 4.25, -2.1), pose_to_tf(20.5, 0.0, 0.0),   # 2: reach leaves the map
                   pose_to_tf(0.0, 0.0, 0.0)])
    tf[3, :3, :3] *= 1.5                                                        # 3: not a rigid transform
    ob._note_ingest(env, tf, reach)
    assert (ob._dirty_obst[2] == full).all() and (ob._dirty_obst[3] == full).all()
    for e in (0, 1):
        # every texel of a worst-case frame (all depths, incl. the corners at max range) lands inside the window
        depth = rng.uniform(0.0, 1.0, (H, W)).astype(np.float32)
        depth[0, 0] = depth[-1, -1] = depth[0, -1] = depth[-1, 0] = 0.999999
        z = depth * (MAX_DEPTH - MIN_DEPTH) + MIN_DEPTH
        cloud = transform_points(tf[e], get_point_cloud(z, z < MAX_DEPTH, fx, fy))
        px = np.rint(cloud[:, :2][:, ::-1] * 20) + S // 2
        col, row = S - px[:, 0], px[:, 1]
        y0, y1, x0, x1 = ob._dirty_obst[e]
        assert row.min() >= y0 and row.max() <= y1 and col.min() >= x0 and col.max() <= x1
        assert (y1 - y0) <= 2 * (int(np.ceil(reach)) + 2) and 0 <= y0 and y1 <= S - 1
    # explore=False (a body camera): navigable is recomputed in the grown window, which stays pending for the reveal
    before = ob._dirty_obst[:2].copy()
    w, nb_nav, _ = ob._take_windows(env[:2], True, False)
    assert (w[:, 0] == before[:, 0] - 3).all() and (w[:, 1] == before[:, 1] + 3).all()
    assert (w[:, 5] < w[:, 4]).all() and (w[:, 9] < w[:, 8]).all()
    assert (ob._dirty_nav[:2] == w[:, :4]).all()
    rows, words = w[:, 1] - w[:, 0] + 1, (w[:, 3] >> 5) - (w[:, 2] >> 5) + 1
    assert nb_nav == -(-int((rows * words).max()) // 256)
    ob._note_ingest(env[:1], np.stack([pose_to_tf(5.0, -2.0, 0.0)]), reach)     # a second camera of slot 0
    w2, _, _ = ob._take_windows(env[:1], True, False)
    box_before = ob._bbox_host[:2].copy()
    w3, _, _ = ob._take_windows(env[:2], False, True, agent[:2], R)             # the reveal without depth
    assert (w3[:, 1] < w3[:, 0]).all() and (w3[:, 4:8] == box_before).all()
    assert w3[0, 8] <= min(w[0, 0], w2[0, 0]) and w3[0, 9] >= max(w[0, 1], w2[0, 1])     # both cameras' windows ...
    assert w3[0, 10] <= min(w[0, 2], w2[0, 2]) and w3[0, 11] >= max(w[0, 3], w2[0, 3])
    assert w3[0, 8] <= ob._bbox_host[0, 0] - 3 and w3[0, 9] >= ob._bbox_host[0, 1] + 3  # ... and the revealed box grown by 3
    assert (ob._dirty_nav[:2, 1] < ob._dirty_nav[:2, 0]).all()

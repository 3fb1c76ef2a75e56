"""-m gpu: the dirty-window forms of navigable_kernel / frontier_prepare_kernel (round 3) against the full-plane forms they
replace (VLFM_FULL_PLANES=1 = NULL windows = the reference's full-map passes, obstacle_map.py:105-109,127,159-163): same
obstacle / navigable / explored planes and the same frontiers at every step, through resets, explore=False calls, reveal-only
calls and a camera whose reach window leaves the map.  (The golden fixtures and the oracle tests run on the windowed form and
pin it against the reference; this file pins it against the other form on call patterns those do not contain.)"""
import numpy as np
import pytest
import torch

from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, camera_intrinsics, depth_frame, pose_to_tf

pytestmark = pytest.mark.gpu
KW = dict(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5)


def _pair(gpu_device, n):
    from vlfm_amd.mapping import ObstacleMapBatch

    a, b = ObstacleMapBatch(n, device=gpu_device, **KW), ObstacleMapBatch(n, device=gpu_device, **KW)
    assert not a.full_planes
    b.full_planes = True
    return a, b


def _same(a, b, where):
    for name in ("obstacle_bits", "navigable_bits", "explored_bits"):
        assert torch.equal(getattr(a, name), getattr(b, name)), (where, name)
    if a.frontiers_ready:
        fa, fb = a.frontiers_px(), b.frontiers_px()
        for e in range(a.n_envs):
            assert np.array_equal(fa[e], fb[e]), (where, e)


def test_windowed_planes_equal_full_planes_over_an_episode_with_resets(gpu_device):
    from vlfm_amd.harness import RoomsRenderer

    E = 8
    fx, fy, fov = camera_intrinsics(640)
    rr = RoomsRenderer(list(range(E)), 500, 480, 640, gpu_device)
    a, b = _pair(gpu_device, E)
    for t in range(70):
        depth, tf = rr.render(t), rr.tf_table[t]
        if t == 30:                       # slots 2 and 5 start a new episode: their first pass is a full-map pass again
            a.reset([2, 5]); b.reset([2, 5])
        for m in (a, b):
            m.ingest(depth, tf, MIN_DEPTH, MAX_DEPTH, fx, fy)
            m.update_after_ingest(tf, MAX_DEPTH, fov)
        _same(a, b, f"step {t}")
    a.check_status(); b.check_status()
    assert bool(a.explored_bits.any())


def test_windowed_planes_equal_full_planes_for_the_robot_call_pattern_and_at_the_map_edge(gpu_device):
    """explore=False per body camera, then a reveal without depth (reality_policies.py:113-141); subsets of slots per call;
    one slot's camera so close to the map edge that its reach window leaves the map (whole-map window)."""
    E = 4
    fx, fy, fov = camera_intrinsics(640)
    rng = np.random.default_rng(11)
    a, b = _pair(gpu_device, E)
    base = [(0.0, 0.0), (3.0, -4.0), (21.0, 0.5), (-2.0, 19.5)]       # slots 2 and 3: windows leave the map
    for t in range(12):
        yaw = 0.5 * t
        for cam in range(2):                                           # two body cameras, obstacles only
            slots = [0, 1, 2, 3] if cam == 0 else [0, 2]               # the second camera exists on two robots only
            depth = torch.from_numpy(np.stack([depth_frame(rng, 480, 640) for _ in slots])).to(gpu_device)
            tf = np.stack([pose_to_tf(base[s][0] + 0.1 * t, base[s][1], yaw + 1.2 * cam) for s in slots])
            for m in (a, b):
                m.ingest(depth, tf, MIN_DEPTH, 2.5, fx, fy, env_ids=slots)
                m.update_after_ingest(tf, 2.5, fov, env_ids=slots, explore=False)
            _same(a, b, f"step {t} camera {cam}")
        slots = [0, 1, 2, 3] if t % 3 else [1, 3]                      # the reveal sometimes skips robots
        tf = np.stack([pose_to_tf(base[s][0] + 0.1 * t, base[s][1], yaw) for s in slots])
        for m in (a, b):
            m.update_after_ingest(tf, 2.5, 2 * fov, env_ids=slots, explore=True, update_obstacles=False)
        _same(a, b, f"step {t} reveal")
    for m in (a, b):
        m.status.zero_()   # slots near the edge may have pushed points off the map: not what this test is about

"""-m gpu: size-independent properties at BASELINE.json's full sizes (configs[3]/[4]: 128 resident 640x480 environments;
configs[4]: 1280x720, 1000x1000 maps, value map synchronised with the explored area), where running the oracle for every
environment would take minutes: batched == single-slot, the fusion algebra's fixed points (SURVEY.md 8c pins), the
explored-area invariant of value_map.py:369-375, monotone obstacle planes, explored within navigable."""
import numpy as np
import pytest
import torch

from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, camera_intrinsics, depth_frame, pose_to_tf

pytestmark = pytest.mark.gpu
KW = dict(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5)


def _frames(E, H, W, steps, seed):
    rng = np.random.default_rng(seed)
    depth = np.stack([np.stack([depth_frame(rng, H, W) for _ in range(E)]) for _ in range(steps)])
    tf = np.stack([np.stack([pose_to_tf(0.25 * t * np.cos(0.3 * e), 0.25 * t * np.sin(0.3 * e), 0.3 * e + 0.5 * t)
                             for e in range(E)]) for t in range(steps)])
    vals = rng.uniform(0.15, 0.45, (steps, E, 1))
    return depth, tf, vals


def test_128_envs_batched_equals_single_slot(gpu_device):
    from vlfm_amd.mapping import ObstacleMap, ObstacleMapBatch, ValueMap, ValueMapBatch

    E, steps = 128, 4
    fx, fy, fov = camera_intrinsics(640)
    depth, tf, vals = _frames(E, 480, 640, steps, 1)
    vb = ValueMapBatch(E, 1, use_max_confidence=False, device=gpu_device)
    ob = ObstacleMapBatch(E, device=gpu_device, **KW)
    probe = [0, 77, 127]
    singles = {e: (ValueMap(1, use_max_confidence=False, device=gpu_device), ObstacleMap(device=gpu_device, **KW)) for e in probe}
    for t in range(steps):
        d = torch.from_numpy(depth[t]).to(gpu_device)
        keys = ob.ingest(d, tf[t], MIN_DEPTH, MAX_DEPTH, fx, fy, want_colmax=True)
        ob.update_after_ingest(tf[t], MAX_DEPTH, fov)
        vb.update(vals[t], None, tf[t], MIN_DEPTH, MAX_DEPTH, fov, colmax=keys)
        for e, (v1, o1) in singles.items():
            o1.update_map(depth[t, e], tf[t, e], MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
            v1.update_map(vals[t, e], depth[t, e], tf[t, e], MIN_DEPTH, MAX_DEPTH, fov)
    fr = ob.frontiers_px()
    obst = ob._unpack(ob.obstacle_bits)
    for e, (v1, o1) in singles.items():
        assert np.array_equal(vb.conf[e].cpu().numpy(), v1._map)            # same kernels, same inputs: bit-identical
        assert np.array_equal(vb.value[e].cpu().numpy(), v1._value_map)
        assert np.array_equal(obst[e].cpu().numpy().astype(bool), o1._map)
        assert np.array_equal(ob.explored[e].cpu().numpy().astype(bool), o1.explored_area)
        assert np.array_equal(fr[e].reshape(-1, 2), np.asarray(o1._frontiers_px, np.float64).reshape(-1, 2))


def test_fusion_fixed_points_at_scale(gpu_device):
    """value_map.py:398-429: first observation -> conf = new, value = values; the same observation again -> unchanged;
    a second value at the same pose (equal confidences) -> the arithmetic mean."""
    from vlfm_amd.mapping import ValueMapBatch

    E = 64
    fov = camera_intrinsics(640)[2]
    depth, tf, vals = _frames(E, 480, 640, 1, 2)
    d = torch.from_numpy(depth[0]).to(gpu_device)
    vb = ValueMapBatch(E, 1, use_max_confidence=False, device=gpu_device)
    vb.update(vals[0], d, tf[0], MIN_DEPTH, MAX_DEPTH, fov)
    c1, v1 = vb.conf.clone(), vb.value.clone()
    seen = c1 > 0
    assert seen.any(dim=(1, 2)).all() and float(c1.max()) <= 1.0   # (edge cells blend with 0 under the bilinear rotate)
    want = torch.from_numpy(vals[0].astype(np.float64)).to(gpu_device).view(E, 1, 1, 1).expand_as(v1)   # w2 == 1.0 exactly
    assert torch.equal(v1[seen], want[seen]) and float(v1[~seen].abs().max()) == 0.0
    vb.update(vals[0], d, tf[0], MIN_DEPTH, MAX_DEPTH, fov)                  # idempotence
    assert torch.allclose(vb.conf, c1, atol=1e-6, rtol=0) and torch.allclose(vb.value, v1, atol=1e-6, rtol=0)
    vb.reset()
    assert float(vb.conf.abs().max()) == 0.0
    vb.update(vals[0], d, tf[0], MIN_DEPTH, MAX_DEPTH, fov)
    other = vals[0] + 0.2
    vb.update(other, d, tf[0], MIN_DEPTH, MAX_DEPTH, fov)
    mean = torch.from_numpy(((vals[0] + other) / 2).astype(np.float64)).to(gpu_device).view(E, 1, 1, 1).expand_as(v1)
    assert torch.allclose(vb.value[seen], mean[seen], atol=1e-6, rtol=0)


def test_config5_hd_sync_invariants(gpu_device):
    """1280x720 depth, 16 environments, value map synchronised with the explored area (value_map.py:369-375)."""
    from vlfm_amd.mapping import ObstacleMapBatch, ValueMapBatch

    E, steps = 16, 5
    fx, fy, fov = camera_intrinsics(1280)
    depth, tf, vals = _frames(E, 720, 1280, steps, 3)
    ob = ObstacleMapBatch(E, device=gpu_device, **KW)
    vb = ValueMapBatch(E, 1, use_max_confidence=False, device=gpu_device, explored_bits=ob.explored_bits)
    prev_obst = None
    for t in range(steps):
        d = torch.from_numpy(depth[t]).to(gpu_device)
        keys = ob.ingest(d, tf[t], MIN_DEPTH, MAX_DEPTH, fx, fy, want_colmax=True)
        ob.update_after_ingest(tf[t], MAX_DEPTH, fov)
        vb.update(vals[t], None, tf[t], MIN_DEPTH, MAX_DEPTH, fov, colmax=keys)
        obst = ob._unpack(ob.obstacle_bits).bool()
        nav = ob._unpack(ob.navigable_bits).bool()
        expl = ob.explored.bool()
        assert not (expl & ~nav).any()                                       # obstacle_map.py:127
        assert not (obst & nav).any()                                        # navigable = ~dilate(obstacles)
        if prev_obst is not None:
            assert not (prev_obst & ~obst).any()                             # obstacle bits are only ever set
        prev_obst = obst
        assert float(vb.conf[~expl].abs().max()) == 0.0                      # the invariant of :369-375
        assert float(vb.value[..., 0][~expl].abs().max()) == 0.0
    assert (vb.conf > 0).any() and bool(ob.explored.bool().any())
    ob.check_status()


def test_config5_hd_sync_16_envs_against_the_oracle(gpu_device):
    """BASELINE configs[4] geometry per GPU for real: 1280x720 depth, 16 environments in ONE batch, value map synchronised
    with the obstacle map's explored area (value_map.py:369-375 -- the full-map mode), 20 steps in the rooms-and-pillars
    world of tests/golden/world500.py (environment e walks the tour from its 30e-th pose); slots 0, 7 and 15 are checked
    against RefValueMap(obstacle_map=RefObstacleMap) fed the same frames: planes and frontiers bit-exact every step,
    confidence / value maps within 1e-4 with identical support."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import world500 as w5
    from oracle.ref_obstacle_map import RefObstacleMap
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ObstacleMapBatch, ValueMapBatch

    E, steps, H, W = 16, 20, 720, 1280
    fx, fy, fov = camera_intrinsics(W)
    probe = [0, 7, 15]
    poses = w5.integrate(w5.plan_actions())
    ob = ObstacleMapBatch(E, device=gpu_device, **KW)
    vb = ValueMapBatch(E, 1, use_max_confidence=False, device=gpu_device, explored_bits=ob.explored_bits)
    refs = {}
    for e in probe:
        rom = RefObstacleMap(**KW)
        refs[e] = (rom, RefValueMap(1, use_max_confidence=False, obstacle_map=rom))
    rng = np.random.default_rng(5)
    worst, n_frontiers = 0.0, 0
    for t in range(steps):
        at = [poses[30 * e + t] for e in range(E)]
        depth = np.stack([w5.depth_from_profile(w5.wall_profile(x, y, k, W), H) for (x, y, k) in at])
        tf = np.stack([w5.tf_of(x, y, k) for (x, y, k) in at])
        vals = rng.uniform(0.15, 0.45, (E, 1))
        d = torch.from_numpy(depth).to(gpu_device)
        keys = ob.ingest(d, tf, MIN_DEPTH, MAX_DEPTH, fx, fy, want_colmax=True)
        ob.update_after_ingest(tf, MAX_DEPTH, fov)
        vb.update(vals, None, tf, MIN_DEPTH, MAX_DEPTH, fov, colmax=keys)
        fr = ob.frontiers_px()
        obst = ob._unpack(ob.obstacle_bits)
        for e in probe:
            rom, rvm = refs[e]
            rom.update_map(depth[e].copy(), tf[e], MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
            rvm.update_map(vals[e], depth[e].copy(), tf[e], MIN_DEPTH, MAX_DEPTH, fov)
            where = f"slot {e} step {t}"
            assert np.array_equal(obst[e].cpu().numpy().astype(bool), rom._map.astype(bool)), where
            assert np.array_equal(ob.explored[e].cpu().numpy().astype(bool), rom.explored_area.astype(bool)), where
            assert np.array_equal(fr[e].reshape(-1, 2), np.asarray(rom._frontiers_px, np.float64).reshape(-1, 2)), where
            conf, val = vb.conf[e].cpu().numpy(), vb.value[e].cpu().numpy()
            assert np.array_equal(conf > 0, rvm._map > 0), where
            ec, ev = np.abs(conf - rvm._map).max(), np.abs(val - rvm._value_map).max()
            assert ec <= 1e-4 and ev <= 1e-4, (where, ec, ev)
            worst = max(worst, float(ec), float(ev))
            n_frontiers += len(rom._frontiers_px)
    ob.check_status()
    assert all(refs[e][0].explored_area.sum() > 3000 for e in probe) and n_frontiers > 60
    print(f"config 5 (16 x 1280x720, explored-area sync): max abs deviation over 20 steps = {worst:.3e}")

"""-m gpu: HIP ValueMap vs the oracle (oracle/ref_value_map.py) on identical seeded inputs, through the C ABI.

BASELINE.json's bar is 1e-4 absolute on the confidence and value maps.  The device holds the confidence map in f32 and the
value map in f64 with the reference's own promotion rule (value_map.py:423), every index- and value-producing operation is
an explicitly rounded IEEE operation, so the bar of `_compare` is EQUALITY (dtype included); TOL remains for the derived
quantities compared elsewhere in this file."""
import numpy as np
import pytest

from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, SyntheticEnv, camera_intrinsics, pose_to_tf

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _fov(width=640):
    return camera_intrinsics(width)[2]


def _compare(ours, ref, tol=TOL):
    c, v = ours._map, ours._value_map
    assert c.shape == ref._map.shape and v.shape == ref._value_map.shape
    assert c.dtype == ref._map.dtype and v.dtype == ref._value_map.dtype, (v.dtype, ref._value_map.dtype)
    ec = np.abs(c - ref._map).max()
    ev = np.abs(v - ref._value_map).max()
    assert np.array_equal(c, ref._map) and np.array_equal(v, ref._value_map), (ec, ev)
    return ec, ev


@pytest.mark.parametrize("use_max_conf", [False, True])
def test_trajectory_parity(gpu_device, use_max_conf):
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMap

    env = SyntheticEnv(3)
    ours = ValueMap(1, use_max_confidence=use_max_conf, device=gpu_device)
    ref = RefValueMap(1, use_max_confidence=use_max_conf)
    for step in range(40):
        depth, tf, values = env.observe()
        ours.update_map(values, depth, tf, MIN_DEPTH, MAX_DEPTH, _fov())
        ref.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, _fov())
        if step % 13 == 0:
            _compare(ours, ref)
    ec, ev = _compare(ours, ref)
    print("max err conf/value", ec, ev)
    # np.median runs in the array's dtype: f64 in the weighted mode, f32 (mean of the two middle elements rounded to f32)
    # where the reference's array stays f32 -- values, their dtype and the permutation all equal
    wps = np.random.default_rng(17).uniform(-5, 5, size=(24, 2))
    s_o, v_o = ours.sort_waypoints(wps, 0.5)
    s_r, v_r = ref.sort_waypoints(wps, 0.5)
    assert np.array_equal(s_o, s_r) and np.array_equal(np.array(v_o, float), np.array(v_r, float))


@pytest.mark.parametrize("fusion", ["replace", "equal_weighting"])
def test_fusion_ablations(gpu_device, fusion):
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMap

    env = SyntheticEnv(5)
    ours = ValueMap(1, use_max_confidence=False, fusion_type=fusion, device=gpu_device)
    ref = RefValueMap(1, use_max_confidence=False, fusion_type=fusion)
    for _ in range(15):
        depth, tf, values = env.observe()
        ours.update_map(values, depth, tf, MIN_DEPTH, MAX_DEPTH, _fov())
        ref.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, _fov())
    _compare(ours, ref)


def test_multichannel_and_arbitrary_yaw(gpu_device):
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMap

    rng = np.random.default_rng(7)
    env = SyntheticEnv(11, channels=2)
    ours = ValueMap(2, use_max_confidence=False, device=gpu_device)
    ref = RefValueMap(2, use_max_confidence=False)
    for k in range(25):
        depth, _, values = env.observe()
        tf = pose_to_tf(rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-np.pi, np.pi))
        ours.update_map(values, depth, tf, MIN_DEPTH, MAX_DEPTH, _fov())
        ref.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, _fov())
    _compare(ours, ref)


def test_depth_extremes_and_channel_dim(gpu_device):
    """depth == 1 everywhere -> whole cone visible; depth == 0 -> profile at min_depth (SURVEY 8c pin 3);
    (H,W,1) depth is squeezed (value_map.py:231-232)."""
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMap

    for fill in (1.0, 0.0, 0.37):
        ours = ValueMap(1, use_max_confidence=False, device=gpu_device)
        ref = RefValueMap(1, use_max_confidence=False)
        depth = np.full((480, 640, 1), fill, np.float32)
        tf = pose_to_tf(1.03, -2.2, 0.7)
        ours.update_map(np.array([0.4]), depth, tf, MIN_DEPTH, MAX_DEPTH, _fov())
        ref.update_map(np.array([0.4]), depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, _fov())
        _compare(ours, ref)


def test_window_clipped_at_map_edge_and_outside_assert(gpu_device):
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMap

    ours = ValueMap(1, use_max_confidence=False, device=gpu_device)
    ref = RefValueMap(1, use_max_confidence=False)
    depth = SyntheticEnv(2).observe()[0]
    for (x, y, yaw) in [(24.9, 24.9, 0.3), (-24.9, 24.0, 2.0), (24.0, -24.95, -1.0), (-24.95, -24.95, 3.0)]:
        tf = pose_to_tf(x, y, yaw)
        ours.update_map(np.array([0.3]), depth, tf, MIN_DEPTH, MAX_DEPTH, _fov())
        ref.update_map(np.array([0.3]), depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, _fov())
    _compare(ours, ref)
    with pytest.raises(AssertionError, match="outside the image"):
        ours.update_map(np.array([0.3]), depth, pose_to_tf(25.2, 0, 0), MIN_DEPTH, MAX_DEPTH, _fov())
    with pytest.raises(AssertionError, match="Incorrect number of values"):
        ours.update_map(np.array([0.3, 0.1]), depth, pose_to_tf(0, 0, 0), MIN_DEPTH, MAX_DEPTH, _fov())


def test_sort_waypoints_parity(gpu_device):
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMap

    env = SyntheticEnv(9)
    ours = ValueMap(1, use_max_confidence=False, device=gpu_device)
    ref = RefValueMap(1, use_max_confidence=False)
    for _ in range(20):
        depth, tf, values = env.observe()
        ours.update_map(values, depth, tf, MIN_DEPTH, MAX_DEPTH, _fov())
        ref.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, _fov())
    rng = np.random.default_rng(3)
    wps = np.concatenate([rng.uniform(-4, 4, size=(12, 2)), np.array([[20.0, 20.0], [-24.99, 24.99], [0.0, 0.0]])])
    s_o, v_o = ours.sort_waypoints(wps, 0.5)
    s_r, v_r = ref.sort_waypoints(wps, 0.5)
    assert np.array_equal(s_o, s_r)  # bit-exact frontier order
    assert np.array_equal(np.array(v_o, float), np.array(v_r, float))  # f64 medians of the f64 map: exact
    assert v_o[-1] == -1  # never-seen waypoint (img_utils.py:257-258)


def test_batched_envs_match_single(gpu_device):
    """One launch for 6 envs == 6 independent oracles (no cross-env leakage)."""
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMapBatch

    E = 6
    envs = [SyntheticEnv(20 + e) for e in range(E)]
    batch = ValueMapBatch(E, 1, use_max_confidence=False, device=gpu_device)
    refs = [RefValueMap(1, use_max_confidence=False) for _ in range(E)]
    for _ in range(10):
        obs = [e.observe() for e in envs]
        depth = np.stack([o[0] for o in obs])
        tf = np.stack([o[1] for o in obs])
        vals = np.stack([o[2] for o in obs])
        batch.update(vals, depth, tf, MIN_DEPTH, MAX_DEPTH, _fov())
        for r, o in zip(refs, obs):
            r.update_map(o[2], o[0].copy(), o[1], MIN_DEPTH, MAX_DEPTH, _fov())
    conf = batch.conf.cpu().numpy()
    val = batch.value.cpu().numpy()
    for e in range(E):
        assert np.array_equal(conf[e], refs[e]._map) and np.array_equal(val[e], refs[e]._value_map)


def test_hd_depth_1280x720(gpu_device):
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMap

    env = SyntheticEnv(4, height=720, width=1280)
    ours = ValueMap(1, use_max_confidence=False, device=gpu_device)
    ref = RefValueMap(1, use_max_confidence=False)
    for _ in range(6):
        depth, tf, values = env.observe()
        ours.update_map(values, depth, tf, MIN_DEPTH, MAX_DEPTH, _fov(1280))
        ref.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, _fov(1280))
    _compare(ours, ref)


def test_record_and_replay_in_the_reference_format(gpu_device, tmp_path, monkeypatch):
    """RECORD_VALUE_MAP / replay (value_map.py:26-30,77-94,130-144,448-475): 8-bit depth PNGs + data.json + kwargs.json.
    A replay must rebuild the map the recording run would have built from the SAME quantised depth."""
    import json

    from oracle.ref_value_map import RefValueMap
    from PIL import Image
    from vlfm_amd.mapping import value_map as vm_mod

    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(vm_mod, "RECORDING", True)
    env = SyntheticEnv(11)
    rec = vm_mod.ValueMap(1, use_max_confidence=False, device=gpu_device)
    frames = [env.observe() for _ in range(5)]
    for depth, tf, values in frames:
        rec.update_map(values, depth, tf, MIN_DEPTH, MAX_DEPTH, _fov())
    monkeypatch.setattr(vm_mod, "RECORDING", False)
    data = json.load(open(tmp_path / "value_map_recordings" / "data.json"))
    assert len(data) == 5 and json.load(open(tmp_path / "value_map_recordings" / "kwargs.json")) == {
        "value_channels": 1, "size": 1000, "use_max_confidence": False}
    replayed = vm_mod.replay_from_dir(device=gpu_device)
    ref = RefValueMap(1, use_max_confidence=False)
    for k, (depth, tf, values) in enumerate(frames):
        png = np.asarray(Image.open(tmp_path / "value_map_recordings" / f"{k:04d}.png")).astype(np.float32) / 255.0
        assert np.array_equal(png, (depth * 255).astype(np.uint8).astype(np.float32) / 255.0)
        ref.update_map(values, png, tf, MIN_DEPTH, MAX_DEPTH, _fov())
    _compare(replayed, ref)


def test_multi_camera_same_slot_is_sequential_and_batch_rejects_duplicates(gpu_device):
    """RealityMixin feeds several cameras into ONE map per step (reality_policies.py:113-141), sequentially.  The drop-in
    does the same; the batched entry point refuses two observations of one slot in a single launch."""
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMap, ValueMapBatch

    env = SyntheticEnv(12)
    ours = ValueMap(1, use_max_confidence=False, device=gpu_device)
    ref = RefValueMap(1, use_max_confidence=False)
    for step in range(4):
        for cam in range(3):   # three cameras, same position, different headings
            depth, _, values = env.observe()
            tf = pose_to_tf(0.3 * step, 0.1 * step, 2.0 * cam + 0.2 * step)
            ours.update_map(values, depth, tf, MIN_DEPTH, MAX_DEPTH, _fov())
            ref.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, _fov())
    _compare(ours, ref)
    vb = ValueMapBatch(2, 1, device=gpu_device)
    d = np.stack([env.observe()[0] for _ in range(2)])
    tfs = np.stack([pose_to_tf(0, 0, 0), pose_to_tf(0, 0, 1)])
    with pytest.raises(AssertionError, match="one observation per environment slot"):
        vb.update(np.array([[0.3], [0.4]]), d, tfs, MIN_DEPTH, MAX_DEPTH, _fov(), env_ids=[1, 1])
    with pytest.raises(AssertionError, match="out of range"):
        vb.update(np.array([[0.3]]), d[:1], tfs[:1], MIN_DEPTH, MAX_DEPTH, _fov(), env_ids=[2])


@pytest.mark.parametrize("size,hfov_deg,max_depth,hw", [(500, 60.0, 3.5, (240, 320)), (700, 90.0, 8.0, (480, 640))])
def test_other_map_sizes_and_camera_models(gpu_device, size, hfov_deg, max_depth, hw):
    """Nothing is specialised to S = 1000 / fov = 79 deg / 5 m: other map sizes (tail words of the bit planes), cone template
    sizes (T = 2 int(max_depth ppm) + 1) and image shapes against the oracle, both maps, with sort_waypoints."""
    from oracle.ref_obstacle_map import RefObstacleMap
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ObstacleMap, ValueMap
    from vlfm_amd.synthetic import depth_frame

    H, W = hw
    fx, fy, fov = camera_intrinsics(W, hfov_deg)
    rng = np.random.default_rng(size)
    kw = dict(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5, size=size)
    ours_o, ref_o = ObstacleMap(device=gpu_device, **kw), RefObstacleMap(**kw)
    ours_v = ValueMap(1, size=size, use_max_confidence=False, device=gpu_device)
    ref_v = RefValueMap(1, size=size, use_max_confidence=False)
    for t in range(8):
        depth = depth_frame(rng, H, W)
        tf = pose_to_tf(0.3 * t, -0.2 * t, 0.6 * t)
        vals = rng.uniform(0.15, 0.45, 1)
        for o in (ours_o, ref_o):
            o.update_map(depth.copy(), tf, MIN_DEPTH, max_depth, fx, fy, fov)
        ours_v.update_map(vals, depth, tf, MIN_DEPTH, max_depth, fov)
        ref_v.update_map(vals, depth.copy(), tf, MIN_DEPTH, max_depth, fov)
        assert np.array_equal(ours_o._map, ref_o._map) and np.array_equal(ours_o.explored_area, ref_o.explored_area), t
        assert np.array_equal(np.asarray(ours_o._frontiers_px, np.float64).reshape(-1, 2),
                              np.asarray(ref_o._frontiers_px, np.float64).reshape(-1, 2)), t
    _compare(ours_v, ref_v)
    if len(ref_o.frontiers):
        a, av = ours_v.sort_waypoints(ref_o.frontiers, 0.5)
        b, bv = ref_v.sort_waypoints(ref_o.frontiers, 0.5)
        assert np.array_equal(a, b) and np.array_equal(np.asarray(av, float), np.asarray(bv, float))


def test_wide_cone_overflows_the_cell_list_and_is_fused_in_place(gpu_device):
    """Round 5: the update collects the cells that receive a confidence in an LDS list of at most 8 192 entries and fuses the list
    one cell per lane.  A 79-degree cone at 5 m marks ~3 000-6 900 cells; a 120-degree cone with nothing in the way marks ~10 500: the
    list overflows and the rest must be fused in place -- same maps as the oracle, every yaw."""
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMap

    H, W = 240, 320
    fov = camera_intrinsics(W, 120.0)[2]
    ours = ValueMap(1, use_max_confidence=False, device=gpu_device)
    ref = RefValueMap(1, use_max_confidence=False)
    rng = np.random.default_rng(8)
    for t in range(7):
        depth = np.ones((H, W), np.float32) if t % 3 else rng.uniform(0.6, 1.0, (H, W)).astype(np.float32)
        tf = pose_to_tf(0.4 * t, 0.3 * t, 0.9 * t - 2.0)
        vals = rng.uniform(0.15, 0.45, 1)
        ours.update_map(vals, depth, tf, MIN_DEPTH, MAX_DEPTH, fov)
        ref.update_map(vals, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fov)
    assert int((ref._map != 0).sum()) > 9000          # (the union of the cones: far more than the list holds per observation)
    _compare(ours, ref)


def test_block_sparse_sweep_at_every_batch_size(gpu_device):
    """Round 5: with at most two workgroups per observation (>= 128 observations on a 256-CU device) the update sweeps only the
    4 x 4 blocks of the window whose source footprint holds a visible bit (csrc/value_map.hip, step 3a / 3b); smaller batches keep
    the tile sweep.  VLFM_VM_TARGET_WGS=1 (read once per process) makes a one-observation launch take the block-sparse path: the
    oracle comparisons of this file and the reference fixtures are repeated under it in a child process."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VLFM_VM_TARGET_WGS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_value_map_gpu.py"), os.path.join(root, "tests", "test_golden_gpu.py"),
                        "-k", "not block_sparse_sweep and not three_launch and not obstacle_map_matches and not multicamera_obstacle "
                              "and not depth_islands"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout, r.stdout[-500:]


def test_random_jagged_profiles_and_angles_against_the_oracle(gpu_device):
    """raster.h against the oracle on RANDOM polygons and angles, not only on the smooth depth profiles of trajectories.
    384 cases in batches of 64: per case a jagged column-maximum profile (independent per column, steps, single-column spikes, all-near,
    all-far: 642-vertex polygons with edges of every slope, 490 000 random edges in total), a yaw drawn from [0, 2 pi) -- a sixth of them
    exact or one-ulp-off multiples of 90 degrees -- and a position whose coordinates sit on and next to the truncation boundaries of
    value_map.py:309-313.  Two observations per case (the second one exercises the weighted fuse on top of the first): confidence map,
    value map and their dtypes equal to the oracle's."""
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMapBatch

    rng = np.random.default_rng(2026)
    E, rounds, H, W = 64, 6, 48, 640          # (the update only sees the column maxima: 48 rows carry the same profiles as 480)
    for rnd in range(rounds):
        batch = ValueMapBatch(E, 1, use_max_confidence=False, device=gpu_device)
        refs = [RefValueMap(1, use_max_confidence=False) for _ in range(E)]
        for obs in range(2):
            depth = np.empty((E, H, W), np.float32)
            tf = np.empty((E, 4, 4), np.float64)
            vals = rng.uniform(0.05, 0.6, size=(E, 1))
            for e in range(E):
                kind = (e + rnd) % 6
                if kind == 0:
                    prof = rng.uniform(0.0, 1.0, W)
                elif kind == 1:
                    prof = np.repeat(rng.uniform(0.0, 1.0, W // 16), 16)
                elif kind == 2:
                    prof = np.full(W, rng.uniform(0.1, 0.9))
                    prof[rng.integers(0, W, 12)] = rng.uniform(0.0, 1.0, 12)
                elif kind == 3:
                    prof = np.clip(np.cumsum(rng.normal(0, 0.03, W)) + 0.5, 0.0, 1.0)
                elif kind == 4:
                    prof = np.full(W, float(rng.choice([0.0, 1.0, 0.5, 1.0 / 3.0])))
                else:
                    prof = np.where(rng.uniform(size=W) < 0.5, 0.02, 0.98)
                d = rng.uniform(0.0, 1.0, (H, W)).astype(np.float32) * prof[None, :].astype(np.float32)
                d[rng.integers(0, H), :] = prof.astype(np.float32)       # the column maxima ARE the profile
                depth[e] = d
                yaw = rng.uniform(0.0, 2 * np.pi)
                if e % 6 == 5:
                    yaw = float(np.nextafter((e // 6 % 4) * np.pi / 2, [np.inf, -np.inf, 0.0][rnd % 3]))
                x, y = rng.uniform(-15, 15, 2)
                if e % 4 == 1:                                            # on / next to a cell boundary of the truncating placement
                    x = np.round(x * 20) / 20 + float(rng.choice([0.0, 1e-12, -1e-12]))
                    y = np.round(y * 20) / 20 + float(rng.choice([0.0, 1e-12, -1e-12]))
                tf[e] = pose_to_tf(x, y, yaw)
            batch.update(vals, depth, tf, MIN_DEPTH, MAX_DEPTH, _fov())
            for e in range(E):
                refs[e].update_map(vals[e], depth[e].copy(), tf[e], MIN_DEPTH, MAX_DEPTH, _fov())
        conf = batch.conf.cpu().numpy()
        val = batch.value.cpu().numpy()
        for e in range(E):
            assert conf[e].dtype == refs[e]._map.dtype
            assert np.array_equal(conf[e], refs[e]._map), (rnd, e, float(np.abs(conf[e] - refs[e]._map).max()))
            assert np.array_equal(val[e].reshape(refs[e]._value_map.shape), refs[e]._value_map), (rnd, e)


def test_random_channels_modes_corner_maps_and_sort_waypoints_against_the_oracle(gpu_device):
    """Random maps in every fusion mode with 1-3 channels -- some hugging a map corner, so that the window is clipped and the disc of a
    waypoint leaves the map -- then random waypoints at several radii: maps, `sort_waypoints` ORDER and values equal the oracle's
    (scratch run over 240 such rounds: profiles/r06_random_parity_stress.txt)."""
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ValueMap

    modes = [("default", False), ("default", True), ("equal_weighting", False), ("replace", False)]
    for rnd in range(24):
        rng = np.random.default_rng(7000 + rnd)
        fusion, use_max = modes[rnd % 4]
        C = 1 + (rnd // 4) % 3
        ours = ValueMap(C, use_max_confidence=use_max, fusion_type=fusion, device=gpu_device)
        ref = RefValueMap(C, use_max_confidence=use_max, fusion_type=fusion)
        centre = rng.uniform(-20, 20, 2) if rnd % 5 else np.array([24.2, -24.3]) * rng.choice([-1, 1], 2)
        for _ in range(int(rng.integers(2, 7))):
            W = 640
            k = int(rng.integers(0, 4))
            prof = (rng.uniform(0, 1, W) if k == 0 else np.repeat(rng.uniform(0, 1, W // 16), 16) if k == 1 else
                    np.clip(np.cumsum(rng.normal(0, 0.04, W)) + rng.uniform(0.2, 0.8), 0, 1) if k == 2 else np.full(W, rng.uniform(0, 1)))
            d = rng.uniform(0, 1, (16, W)).astype(np.float32) * prof[None].astype(np.float32)
            d[0] = prof.astype(np.float32)
            p = centre + rng.uniform(-0.6, 0.6, 2)
            tf = pose_to_tf(p[0], p[1], rng.uniform(-np.pi, np.pi))
            vals = rng.uniform(0.0, 0.6, C)
            ref.update_map(vals, d.copy(), tf, MIN_DEPTH, MAX_DEPTH, _fov())
            ours.update_map(vals, d.copy(), tf, MIN_DEPTH, MAX_DEPTH, _fov())
        assert np.array_equal(ours._map, ref._map), rnd
        assert np.asarray(ours._value_map).dtype == np.asarray(ref._value_map).dtype, rnd
        assert np.array_equal(np.asarray(ours._value_map), np.asarray(ref._value_map)), rnd
        wps = np.clip(centre + rng.uniform(-6, 6, (int(rng.integers(1, 24)), 2)), -24.4, 24.4)
        radius = float(rng.choice([0.5, 0.25, 1.0, 0.05]))
        kw = dict(reduce_fn=(lambda vs: [max(v) for v in vs])) if C > 1 else {}
        a, b = ours.sort_waypoints(wps, radius, **kw), ref.sort_waypoints(wps, radius, **kw)
        assert np.array_equal(a[0], b[0]), rnd
        assert np.array_equal(np.array(a[1], float), np.array(b[1], float)), rnd


@pytest.mark.parametrize("seed", range(4))
def test_explored_synchronised_mode_on_random_clutter_against_the_oracle(gpu_device, seed):
    """ValueMap(obstacle_map=...) (value_map.py:369-375: new data, confidences and values are cleared wherever the obstacle map's
    explored area is 0) with RANDOM clutter, headings and jumps, one fusion mode per seed: after every step the GPU pair's explored
    area, confidence map and value map equal the oracle pair's.  (scratch run over 60 seeds: profiles/r06_random_parity_stress.txt)"""
    from oracle.ref_obstacle_map import RefObstacleMap
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.mapping import ObstacleMap, ValueMap
    from vlfm_amd.synthetic import depth_frame

    fx, fy, fov = camera_intrinsics(640)
    rng = np.random.default_rng(40_000 + seed)
    fusion, use_max = [("default", False), ("default", True), ("equal_weighting", False), ("replace", False)][seed % 4]
    kw = dict(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5)
    om, rom = ObstacleMap(device=gpu_device, **kw), RefObstacleMap(**kw)
    vm = ValueMap(1, use_max_confidence=use_max, fusion_type=fusion, obstacle_map=om, device=gpu_device)
    rvm = RefValueMap(1, use_max_confidence=use_max, fusion_type=fusion, obstacle_map=rom)
    x = y = 0.0
    cleared = 0
    for step in range(30):
        yaw = rng.uniform(-np.pi, np.pi) if step % 3 else float(int(rng.integers(-6, 7)) * np.pi / 6)
        x += rng.uniform(-0.5, 0.5)
        y += rng.uniform(-0.5, 0.5)
        d = depth_frame(rng)
        if step % 3 != 2:
            d[:] = np.maximum(d, np.float32(0.85))
        for _ in range(int(rng.integers(0, 6))):
            c0 = int(rng.integers(0, 600)); w = int(rng.integers(4, 120)); r0 = int(rng.integers(0, 300)); h = int(rng.integers(40, 480 - r0))
            d[r0:r0 + h, c0:c0 + w] = np.float32(rng.uniform(0.05, 0.7))
        tf = pose_to_tf(x, y, yaw)
        vals = rng.uniform(0.05, 0.6, 1)
        before = int((rvm._map > 0).sum())
        for o, v in ((om, vm), (rom, rvm)):
            o.update_map(d.copy(), tf, MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
            v.update_map(vals, d.copy(), tf, MIN_DEPTH, MAX_DEPTH, fov)
        if not np.array_equal(om.explored_area, rom.explored_area):
            pytest.skip("an extreme-angle tie of the reference's fog of war (platform-dependent: tests/test_obstacle_map_gpu.py)")
        assert np.array_equal(vm._map, rvm._map), (seed, step)
        assert np.array_equal(np.asarray(vm._value_map), np.asarray(rvm._value_map)), (seed, step)
        cleared += int((rvm._map > 0).sum()) < before
    assert (rvm._map > 0).sum() > 500

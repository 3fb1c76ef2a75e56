"""-m gpu: HIP ObstacleMap vs the oracle (oracle/ref_obstacle_map.py) on identical inputs, through the C ABI.
Bar: bit-exact obstacle / navigable / explored planes and bit-exact frontier pixel coordinates (f64)."""
import numpy as np
import pytest

from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, SyntheticEnv, camera_intrinsics, depth_frame, pose_to_tf

pytestmark = pytest.mark.gpu
FX, FY, FOV = camera_intrinsics(640)


def _pair(gpu_device, **kw):
    from oracle.ref_obstacle_map import RefObstacleMap
    from vlfm_amd.mapping import ObstacleMap

    args = dict(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5)
    args.update(kw)
    return ObstacleMap(device=gpu_device, **args), RefObstacleMap(**args)


def _same(ours, ref, step=None):
    assert np.array_equal(ours._map, ref._map), f"obstacle plane differs at step {step}"
    assert np.array_equal(ours._navigable_map != 0, np.asarray(ref._navigable_map) != 0), f"navigable differs at {step}"
    a, b = ours.explored_area, ref.explored_area
    assert np.array_equal(a, b), f"explored differs at step {step}: {np.argwhere(a != b)[:5]}, {(a != b).sum()} cells"
    fo, fr = np.asarray(ours._frontiers_px), np.asarray(ref._frontiers_px)
    assert fo.shape == fr.shape, f"frontier count differs at step {step}: {fo} vs {fr}"
    assert np.array_equal(fo, fr), f"frontier pixels differ at step {step}: {fo} vs {fr}"
    assert np.array_equal(np.asarray(ours.frontiers), np.asarray(ref.frontiers))


@pytest.mark.parametrize("env_id", [3, 8])
def test_trajectory_parity(gpu_device, env_id):
    ours, ref = _pair(gpu_device)
    env = SyntheticEnv(env_id)
    for step in range(30):
        depth, tf, _ = env.observe()
        ours.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
        ref.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
        _same(ours, ref, step)
    assert ref.explored_area.sum() > 1000 and ref._map.sum() > 100


def _room_depth(rng, near=2.0, far=4.8):
    """A frame whose wall is near on one side and beyond max range on the other: leaves open frontiers."""
    d = depth_frame(rng)
    d[:, :320] = np.minimum(d[:, :320], np.float32((near - MIN_DEPTH) / (MAX_DEPTH - MIN_DEPTH)))
    d[:, 320:] = 1.0
    return d


def test_open_space_frontiers_and_arbitrary_poses(gpu_device):
    ours, ref = _pair(gpu_device)
    rng = np.random.default_rng(0)
    pose_rng = np.random.default_rng(1)
    x = y = 0.0
    n_with_frontiers = 0
    for step in range(25):
        yaw = pose_rng.uniform(-np.pi, np.pi)
        x += pose_rng.uniform(-0.3, 0.3)
        y += pose_rng.uniform(-0.3, 0.3)
        tf = pose_to_tf(x, y, yaw)
        depth = _room_depth(rng)
        ours.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
        ref.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
        _same(ours, ref, step)
        n_with_frontiers += len(ref.frontiers) > 0
    assert n_with_frontiers >= 5


def test_no_obstacle_in_cone_reveals_nothing(gpu_device):
    """reveal_fog_of_war returns the fog unchanged when the cone holds no obstacle."""
    ours, ref = _pair(gpu_device)
    depth = np.ones((480, 640), np.float32)
    tf = pose_to_tf(0.0, 0.0, 0.3)
    ours.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
    ref.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
    _same(ours, ref)
    assert ref.explored_area.sum() == 0 and len(ref.frontiers) == 0


def test_update_flags_and_reset(gpu_device):
    ours, ref = _pair(gpu_device)
    env = SyntheticEnv(5)
    for step in range(6):
        depth, tf, _ = env.observe()
        kw = dict(explore=step % 2 == 0, update_obstacles=step % 3 != 0)
        ours.update_map(depth if kw["update_obstacles"] else None, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV, **kw)
        ref.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV, **kw)
        assert np.array_equal(ours._map, ref._map) and np.array_equal(ours.explored_area, ref.explored_area)
    ours.reset()
    ref.reset()
    depth, tf, _ = env.observe()
    ours.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
    ref.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
    _same(ours, ref)


def test_map_edge_index_error_and_hole_flag(gpu_device):
    ours, ref = _pair(gpu_device)
    depth = SyntheticEnv(1).observe()[0]
    tf = pose_to_tf(24.0, 0.0, 0.0)  # looking at the +x edge from 1 m away: points fall beyond row 999
    with pytest.raises(IndexError):
        ref.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
    with pytest.raises(IndexError):
        ours.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
    # hole_area_thresh == -1: zeros are filled with 1.0 and vanish beyond max_depth (obstacle_map.py:87-89)
    o2, r2 = _pair(gpu_device, hole_area_thresh=-1)
    d2 = SyntheticEnv(2, holes=True).observe()[0]
    assert (d2 == 0).any()
    tf2 = pose_to_tf(0.0, 0.0, 0.0)
    o2.update_map(d2, tf2, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
    r2.update_map(d2, tf2, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
    _same(o2, r2)


def _hole_cases():
    """Depth images with invalid (zero) texels: rectangles, speckle, a ring enclosing a valid island, a hole larger than
    the area threshold, holes touching the image border, single pixels and diagonal (8-connected) chains."""
    rng = np.random.default_rng(77)
    base = SyntheticEnv(9).observe()[0]
    cases = {}
    d = base.copy(); d[100:140, 200:260] = 0; d[300:310, 50:55] = 0
    cases["rects"] = d
    d = base.copy(); d[rng.random(d.shape) < 0.02] = 0
    cases["speckle"] = d
    d = base.copy(); d[150:260, 250:400] = 0; d[180:230, 290:360] = base[180:230, 290:360]; d[200:210, 310:330] = 0
    cases["ring_island_nested"] = d
    d = base.copy(); d[40:400, 100:500] = 0  # 360 x 400 = 144000 px > 100000: stays a hole (z = min_depth points)
    cases["big"] = d
    d = base.copy(); d[0:30, 0:40] = 0; d[470:480, 600:640] = 0; d[0:5, 300:340] = 0; d[200:260, 636:640] = 0
    cases["borders"] = d
    d = base.copy()
    for k in range(40):
        d[250 + k, 100 + k] = 0           # diagonal chain: one 8-connected component
    d[50, 50] = 0; d[52, 52] = 0; d[300, 639] = 0; d[479, 0] = 0
    cases["diag_single"] = d
    d = base.copy(); d[:] = 0
    cases["all_zero"] = d
    return cases


@pytest.mark.parametrize("thresh", [100000, 600, 40])
def test_fill_small_holes_parity(gpu_device, thresh):
    """img_utils.py:361-390 on the device (depth_holes.hip): the filled-texel plane and the resulting obstacle map against
    the oracle, for every case and an area threshold that fills all / some / almost no contours."""
    from oracle.ref_geometry import fill_small_holes

    tf = pose_to_tf(0.3, -0.2, 0.4)
    for name, depth in _hole_cases().items():
        ours, ref = _pair(gpu_device, hole_area_thresh=thresh)
        ours.update_map(depth, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
        ref.update_map(depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
        # texels the reference rewrites to 1.0 (probe image: 0 in the holes, 0.5 elsewhere)
        want = fill_small_holes(np.where(depth == 0, 0, 0.5).astype(np.float32), thresh) == 1
        got_bits = ours._batch._filled_bits[0].cpu().numpy().view(np.uint32)
        got = ((got_bits[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(depth.shape[0], -1)[:, :depth.shape[1]]
        assert np.array_equal(got.astype(bool), want), f"{name}: filled plane differs in {(got.astype(bool) != want).sum()} texels"
        _same(ours, ref)


def test_fill_small_holes_batched_and_clean_frames(gpu_device):
    """A frame without zeros after a frame with zeros must not inherit the old fill; batch slots are independent."""
    from oracle.ref_obstacle_map import RefObstacleMap
    from vlfm_amd.mapping import ObstacleMapBatch
    import torch

    cases = _hole_cases()
    clean = SyntheticEnv(9).observe()[0]
    frames = [np.stack([cases["rects"], clean, cases["ring_island_nested"]]),
              np.stack([clean, cases["speckle"], clean])]
    kw = dict(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5)
    batch = ObstacleMapBatch(3, device=gpu_device, **kw)
    refs = [RefObstacleMap(**kw) for _ in range(3)]
    tfs = np.stack([pose_to_tf(0.1 * e, 0.0, 0.2 * e) for e in range(3)])
    for f in frames:
        batch.ingest(torch.from_numpy(f).to(gpu_device), tfs, MIN_DEPTH, MAX_DEPTH, FX, FY)
        batch.check_status()
        batch.update_after_ingest(tfs, MAX_DEPTH, FOV)
        for e in range(3):
            refs[e].update_map(f[e].copy(), tfs[e], MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
    obst = batch._unpack(batch.obstacle_bits).cpu().numpy().astype(bool)
    for e in range(3):
        assert np.array_equal(obst[e], refs[e]._map.astype(bool)), f"slot {e}"
        assert np.array_equal(batch.explored[e].cpu().numpy().astype(bool), refs[e].explored_area.astype(bool))


def _world500_frames(n):
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import world500 as w5

    return [(d, tf) for _, d, tf, _ in w5.episode(w5.plan_actions()[:n])]


@pytest.mark.parametrize("caps", [dict(CAP_FRONTIERS=2, READ_FRONTIERS=2), dict(CAP_PTS=96), dict(CAP_CONTOURS=2)])
def test_scratch_capacity_overflow_raises_instead_of_a_silent_wrong_map(gpu_device, caps):
    """CAP_PTS / CAP_CONTOURS / CAP_FRONTIERS are fixed scratch sizes: when a step needs more (here: deliberately tiny
    capacities in a world with a dozen frontiers) the frontier read-back of that step must raise, whichever kernel of the
    explore pipeline (fog of war, component selection, frontier extraction) ran out."""
    import torch

    from vlfm_amd.mapping.obstacle_map import ObstacleMapBatch

    Tiny = type("Tiny", (ObstacleMapBatch,), caps)
    ob = Tiny(1, min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5, device=gpu_device)
    with pytest.raises(RuntimeError, match="capacity"):
        for depth, tf in _world500_frames(40):
            ob.ingest(torch.from_numpy(depth[None]).to(gpu_device), tf[None], MIN_DEPTH, MAX_DEPTH, FX, FY)
            ob.update_after_ingest(tf[None], MAX_DEPTH, FOV)
            ob.frontiers_px()


def test_harness_surfaces_off_map_obstacle_points_at_episode_end(gpu_device):
    """A 10 m x 10 m map (size 200) is too small for 5 m of depth range around a moving agent: the scatter flags the
    reference's IndexError (obstacle_map.py:101); the batched harness must raise it no later than the episode end."""
    from vlfm_amd.harness import BatchedEpisodes

    sim = BatchedEpisodes(2, device=gpu_device, use_blip2=False, map_size=200, episode_len=25, world="random")
    with pytest.raises(IndexError):
        for _ in range(26):
            sim.step()


def _extreme_angle_tie(ref, tf, ulps=8):
    """True when reveal_fog_of_war at this pose has to break a (near-)tie between the angular extremes of a CONVEX obstacle blob:
    two vertices collinear with the agent (the agent in the column / row / diagonal of a blob's edge).  The reference picks the
    extreme by np.argmin / np.argmax over np.arctan2 of rotated integer vectors: for collinear vertices the mathematical angles are
    EQUAL and which one wins is decided by the last bit of the NumPy build's vectorised arctan2 and of its BLAS (this image's NumPy
    separates (0, 3) and (0, 11) by one ulp, glibc's atan2 returns the same double for both) -- the reference's own outcome is
    platform-dependent there, so such poses are not parity cases."""
    from oracle import cv
    from oracle.ref_frontier_exploration import wrap_heading
    from oracle.ref_obstacle_map import yaw_of

    nav = np.asarray(ref._navigable_map).astype(np.uint8)
    cell = ref._xy_to_px(tf[:2, 3].reshape(1, 2))[0].astype(int)
    angle = np.rad2deg(wrap_heading(yaw_of(tf) + np.pi / 2))
    fov = np.rad2deg(FOV)
    r = int(MAX_DEPTH * ref.pixels_per_meter)
    cone = cv.ellipse(np.zeros_like(nav), cell, (r, r), 0, angle - fov / 2, angle + fov / 2, 1, -1)
    blobs, _ = cv.findContours(cv.bitwise_and(cone, 1 - nav), cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    rot = np.array([[np.cos(-angle), -np.sin(-angle)], [np.sin(-angle), np.cos(-angle)]])
    for blob in blobs:
        if not cv.isContourConvex(blob):
            continue
        v = np.matmul(blob.reshape(-1, 2) - cell, rot)
        a = np.sort(np.arctan2(v[:, 1], v[:, 0]))
        if len(a) > 1 and (a[1] - a[0] <= ulps * np.spacing(abs(a[0])) or a[-1] - a[-2] <= ulps * np.spacing(abs(a[-1]))):
            return True
    return False


@pytest.mark.parametrize("seed", range(8))
def test_random_angles_random_clutter_against_the_oracle(gpu_device, seed):
    """the fog-of-war sector (cv2.ellipse polygon at an arbitrary integer-rounded heading), the shadow lines, the
    visible-contour pick, the explored-area selection and the frontier tracing on RANDOM headings and RANDOM clutter instead of the
    smooth tours of the trajectory tests: 40 steps per seed, each with a fresh uniformly random yaw (every fourth an exact or
    one-ulp-off multiple of 45 degrees), a jump of up to 0.6 m, and a depth frame with 0-6 random near boxes (pillars and wall pieces at
    random columns / ranges) in front of a far wall.  The agent regularly stands INSIDE the padding of an obstacle (non-navigable cell)
    -- this test found the signed-zero rule of the shadow-point rotation that way (csrc/obstacle_map.hip).  Obstacles are scattered
    first (explore=False), the reveal follows as its own call at the same pose unless that pose is an extreme-angle tie
    (_extreme_angle_tie: the reference is platform-dependent there), in which case the reveal pose is nudged.  Planes and frontier
    pixels bit-exact after EVERY step."""
    ours, ref = _pair(gpu_device)
    rng = np.random.default_rng(1000 + seed)
    x = y = 0.0
    grown = nudged = 0
    for step in range(40):
        yaw = rng.uniform(-np.pi, np.pi)
        if step % 4 == 3:
            yaw = float(np.nextafter(int(rng.integers(-4, 5)) * np.pi / 4, [np.inf, -np.inf, 0.0][step % 3]))
        x += rng.uniform(-0.6, 0.6)
        y += rng.uniform(-0.6, 0.6)
        d = depth_frame(rng)
        if step % 3 != 2:
            d[:] = np.maximum(d, np.float32(0.85))                        # a far wall ...
        for _ in range(int(rng.integers(0, 7))):                          # ... with near clutter in front of it
            c0 = int(rng.integers(0, 600)); w = int(rng.integers(4, 120))
            r0 = int(rng.integers(0, 300)); h = int(rng.integers(40, 480 - r0))
            d[r0:r0 + h, c0:c0 + w] = np.float32(rng.uniform(0.05, 0.7))
        tf = pose_to_tf(x, y, yaw)
        for m in (ours, ref):
            m.update_map(d, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV, explore=False)
        tf2 = tf
        for _ in range(20):
            if not _extreme_angle_tie(ref, tf2):
                break
            nudged += 1
            tf2 = pose_to_tf(x + rng.uniform(-0.3, 0.3), y + rng.uniform(-0.3, 0.3), yaw)
        else:
            continue
        before = int(ref.explored_area.sum())
        for m in (ours, ref):
            m.update_map(None, tf2, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV, update_obstacles=False)
        _same(ours, ref, step)
        grown += int(ref.explored_area.sum()) > before
    assert grown >= 3 and ref._map.sum() > 200          # the sequence really revealed area and placed obstacles
    # (tools: the same body over seeds 0..199 = 8 000 random steps: no parity difference, profiles/r06_random_parity_stress.txt)


def _random_hole_frame(rng):
    """A depth frame with random ZERO regions: rectangles, discs, rings with valid islands inside, thin cracks, speckle, zero texels
    touching the image border, holes inside holes."""
    d = depth_frame(rng)
    H, W = d.shape
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(int(rng.integers(1, 7))):
        kind = int(rng.integers(0, 6))
        cy, cx = int(rng.integers(0, H)), int(rng.integers(0, W))
        if kind == 0:
            h, w = int(rng.integers(1, 90)), int(rng.integers(1, 120))
            d[cy:cy + h, cx:cx + w] = 0
        elif kind == 1:
            r = int(rng.integers(1, 45))
            d[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0
        elif kind == 2:                                  # ring: a hole with a valid island (img_utils.py:385-388 rewrites it too)
            r = int(rng.integers(6, 50)); t = int(rng.integers(1, 6))
            rr = (yy - cy) ** 2 + (xx - cx) ** 2
            d[(rr <= r * r) & (rr >= (r - t) ** 2)] = 0
        elif kind == 3:                                  # a diagonal crack, one texel wide
            n = int(rng.integers(5, 120)); s = int(rng.choice([-1, 1]))
            ys = np.clip(cy + np.arange(n), 0, H - 1); xs = np.clip(cx + s * np.arange(n), 0, W - 1)
            d[ys, xs] = 0
        elif kind == 4:                                  # speckle
            m = rng.uniform(size=(40, 60)) < 0.3
            y0, x0 = min(cy, H - 40), min(cx, W - 60)
            d[y0:y0 + 40, x0:x0 + 60][m] = 0
        else:                                            # a band along an image border
            side = int(rng.integers(0, 4)); t = int(rng.integers(1, 12))
            if side == 0: d[:t] = 0
            elif side == 1: d[-t:] = 0
            elif side == 2: d[:, :t] = 0
            else: d[:, -t:] = 0
    return d


@pytest.mark.parametrize("seed", range(4))
def test_random_hole_patterns_against_the_oracle(gpu_device, seed):
    """fill_small_holes + the speculative scatter / journal undo / hole scatter of depth_holes.hip and depth_ingest.hip on RANDOM zero
    patterns (rings with islands, cracks, speckle, border bands, overlaps) instead of the six hand-made cases: 12 frames per seed with
    the reference's default threshold, a threshold that fills only some contours and -1; obstacle planes bit-exact after every frame."""
    rng = np.random.default_rng(500 + seed)
    for thresh in (100000, int(rng.integers(30, 4000)), -1):
        ours, ref = _pair(gpu_device, hole_area_thresh=thresh)
        for k in range(4):
            d = _random_hole_frame(rng)
            tf = pose_to_tf(rng.uniform(-2, 2), rng.uniform(-2, 2), rng.uniform(-np.pi, np.pi))
            for m in (ours, ref):
                m.update_map(d.copy(), tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV, explore=False)
            assert np.array_equal(ours._map, ref._map), (seed, thresh, k, int((ours._map != ref._map).sum()))


def test_long_random_walk_with_arbitrary_headings_in_the_rooms_world(gpu_device):
    """The consistent rooms-and-pillars world of vlfm_amd/synthetic.py (the 500-step fixture's) on a RANDOM walk with arbitrary headings:
    the explored area grows to tens of thousands of cells with a dozen frontiers -- the large, ragged outlines the short random-clutter
    sequences never reach.  160 steps here; tools/stress/long_stress.py ran 36 such episodes of 400-500 steps (15 000+ steps, every plane
    and every frontier pixel after every step: profiles/r06_random_parity_stress.txt)."""
    from vlfm_amd import synthetic as syn

    def wall_profile_any(x, y, yaw, width=640):
        fx = syn.camera_intrinsics(width)[0]
        c, s = float(np.cos(yaw)), float(np.sin(yaw))
        m = -(np.arange(width, dtype=np.float64) - width // 2) / fx
        dx, dy = (c - s * m)[:, None], (s + c * m)[:, None]
        dx = np.where(np.abs(dx) < 1e-12, 1e-12, dx)
        dy = np.where(np.abs(dy) < 1e-12, 1e-12, dy)
        B = syn.BOXES
        tx0, tx1 = (B[None, :, 0] - x) / dx, (B[None, :, 2] - x) / dx
        ty0, ty1 = (B[None, :, 1] - y) / dy, (B[None, :, 3] - y) / dy
        tmin = np.maximum(np.minimum(tx0, tx1), np.minimum(ty0, ty1))
        tmax = np.minimum(np.maximum(tx0, tx1), np.maximum(ty0, ty1))
        hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)
        return np.where(hit, tmin, np.inf).min(axis=1).astype(np.float32)

    ours, ref = _pair(gpu_device)
    rng = np.random.default_rng(90_000)
    x = y = 0.0
    for step in range(160):
        yaw = rng.uniform(-np.pi, np.pi) if step % 5 else float(int(rng.integers(-6, 7)) * np.pi / 6)
        for _ in range(30):
            nx, ny = x + rng.uniform(-0.6, 0.6), y + rng.uniform(-0.6, 0.6)
            if abs(nx) < 9.3 and abs(ny) < 9.3 and not syn._blocked(nx, ny, 0.35):
                x, y = float(nx), float(ny)
                break
        d = syn.depth_from_profile(wall_profile_any(x, y, yaw))
        tf = pose_to_tf(x, y, yaw)
        for m in (ours, ref):
            m.update_map(d, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV, explore=False)
        tf2 = tf
        for _ in range(20):
            if not _extreme_angle_tie(ref, tf2):
                break
            tf2 = pose_to_tf(x + rng.uniform(-0.3, 0.3), y + rng.uniform(-0.3, 0.3), yaw)
        else:
            continue
        for m in (ours, ref):
            m.update_map(None, tf2, MIN_DEPTH, MAX_DEPTH, FX, FY, FOV, update_obstacles=False)
        _same(ours, ref, step)
    assert ref.explored_area.sum() > 15000 and len(np.asarray(ref._frontiers_px).reshape(-1, 2)) >= 4

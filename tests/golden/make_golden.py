#!/usr/bin/env python
"""tests/golden/make_golden.py -- generate (or --check) the golden fixtures from THE REFERENCE'S OWN SOURCE.

Runs /root/reference/vlfm/mapping/{value_map,obstacle_map}.py and vlfm/utils/{geometry_utils,img_utils}.py -- the real
files, imported through oracle/ref_shim.py, which stands in for the three absent third-party packages (cv2 ->
oracle/cvport.c restatement; frontier_exploration -> oracle/ref_frontier_exploration.py).  So the fixtures pin the
in-tree arithmetic of the reference (dtype promotions, truncations, fusion algebra, index conventions, control flow);
the OpenCV / frontier_exploration rules stay restatements (PARITY UNPINNED for those, see oracle/ref_shim.py).

    python tests/golden/make_golden.py            # (re)write tests/golden/*.npz   (needs /root/reference)
    python tests/golden/make_golden.py --check    # regenerate in memory and compare with the committed files
    python tests/golden/make_golden.py --check --real-cv2   # the same with the REAL opencv-python (and frontier_exploration, if
                                                  # importable) under the reference instead of the stand-ins: per fixture the
                                                  # largest difference and the first differing cell (tools/verify_with_real_vlfm.md)

Inputs are the deterministic synthetic episodes of vlfm_amd/synthetic.py; each fixture stores the poses
and values explicitly and a SHA-256 of every depth frame, so a consumer that regenerates the depth from the seed can
prove it fed identical bytes.  Maps are stored sparsely (flat index + value of the non-zero cells) to keep the
fixtures small.
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, SyntheticEnv, camera_intrinsics  # noqa: E402

# ------------------------------------------------------------------------------------------------ case table
VM_CASES = {
    # name: (env seed, steps, channels, use_max_confidence, fusion_type, height, width)
    "vm_default_c1": (3, 30, 1, False, "default", 480, 640),
    "vm_maxconf_c1": (4, 16, 1, True, "default", 480, 640),
    "vm_default_c2": (6, 14, 2, False, "default", 480, 640),
    "vm_replace_c1": (5, 8, 1, False, "replace", 480, 640),
    "vm_equal_c1": (5, 8, 1, False, "equal_weighting", 480, 640),
    "vm_default_hd": (8, 6, 1, False, "default", 720, 1280),
}
OM_CASES = {
    # name: (env seed, steps, holes, hole_area_thresh)
    "om_traj": (1, 26, False, 100000),
    "om_holes_fill": (2, 6, True, 100000),
    "om_holes_all": (2, 6, True, -1),
    # the robot deployment's obstacle band and footprint (config/experiments/reality.yaml:19-21): 0.1-1.5 m puts most
    # wall texels in the band, agent_radius 0.2 makes the navigable-map dilation 9x9 instead of 7x7
    "om_reality": (9, 10, True, 100000, dict(min_height=0.1, max_height=1.5, agent_radius=0.2, area_thresh=1.5)),
}
SYNC_CASE = ("vm_sync_explored", 7, 14)  # ValueMap(obstacle_map=...) full-map mode: (name, seed, steps)
OBSTACLE_KW = dict(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def sparse(a: np.ndarray):
    flat = np.ascontiguousarray(a).reshape(-1)
    idx = np.flatnonzero(flat).astype(np.int32)
    return idx, flat[idx]


def packbits(a: np.ndarray) -> np.ndarray:
    return np.packbits(np.asarray(a).astype(bool), axis=None)


def frontier_blob(per_step):
    """list of (F,2) arrays -> (counts, concatenated) so that empty steps survive the round trip."""
    counts = np.array([len(f) for f in per_step], np.int32)
    cat = np.concatenate([np.asarray(f, np.float64).reshape(-1, 2) for f in per_step]) if counts.sum() else \
        np.zeros((0, 2))
    return counts, cat


# ------------------------------------------------------------------------------------------------ generators
def gen_value_map(ref_vm, seed, steps, channels, use_max, fusion, H, W):
    fov = camera_intrinsics(W)[2]
    env = SyntheticEnv(seed, H, W, channels=channels)
    vm = ref_vm.ValueMap(channels, use_max_confidence=use_max, fusion_type=fusion)
    tfs, vals, hashes = [], [], []
    for _ in range(steps):
        depth, tf, values = env.observe()
        hashes.append(sha(depth))
        tfs.append(tf)
        vals.append(values)
        vm.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fov)
    ci, cv = sparse(vm._map)
    vi, vv = sparse(vm._value_map)
    # frontier scoring on the final map: a ring of waypoints around the last pose + two never-seen ones
    x, y = tfs[-1][0, 3], tfs[-1][1, 3]
    ang = np.linspace(0, 2 * np.pi, 9, endpoint=False)
    wps = np.concatenate([np.stack([x + 1.2 * np.cos(ang), y + 1.2 * np.sin(ang)], 1), [[20.0, 20.0], [-18.5, 3.25]]])
    out = dict(seed=seed, steps=steps, channels=channels, use_max_confidence=use_max, fusion_type=fusion, height=H,
               width=W, min_depth=MIN_DEPTH, max_depth=MAX_DEPTH, fov=fov, tf=np.stack(tfs), values=np.stack(vals),
               depth_sha256=np.array(hashes), conf_idx=ci, conf_val=cv.astype(np.float32), value_idx=vi,
               value_val=np.asarray(vv, np.float64), value_dtype=str(vm._value_map.dtype), waypoints=wps)
    if channels == 1:
        s_wp, s_val = vm.sort_waypoints(wps, 0.5)
        out.update(sorted_waypoints=np.asarray(s_wp), sorted_values=np.asarray(s_val, np.float64))
    else:
        s_wp, s_val = vm.sort_waypoints(wps, 0.5, reduce_fn=lambda vs: [max(v) for v in vs])
        out.update(sorted_waypoints=np.asarray(s_wp), sorted_values=np.asarray(s_val, np.float64))
    return out


def two_camera_script(seed: int, steps: int):
    """Per step two value-map observations into the SAME map with different optics: the 79-degree / 5 m camera and a
    narrower 60-degree camera that is only trusted to 2.5 m, yawed 40 degrees to the left (itm_policy.py:204-206 loops over
    value_map_rgbd; the reference keeps one confidence-cone cache entry per (fov, max_depth), value_map.py:37,339-353)."""
    from vlfm_amd.synthetic import depth_frame, pose_to_tf

    env = SyntheticEnv(seed)
    for _ in range(steps):
        x, y, yaw = env.traj.step()
        yield [(depth_frame(env.rng, 480, 640), pose_to_tf(x, y, yaw), MIN_DEPTH, MAX_DEPTH, camera_intrinsics(640)[2],
                env.rng.uniform(0.15, 0.45, size=1)),
               (depth_frame(env.rng, 480, 640), pose_to_tf(x, y, yaw + 0.7), MIN_DEPTH, 2.5, np.deg2rad(60.0),
                env.rng.uniform(0.15, 0.45, size=1))]


def gen_two_cameras(ref_vm, seed=16, steps=10):
    vm = ref_vm.ValueMap(1, use_max_confidence=False)
    hashes = []
    for cams in two_camera_script(seed, steps):
        for depth, tf, lo, hi, fov, values in cams:
            hashes.append(sha(depth))
            vm.update_map(values, depth.copy(), tf, lo, hi, fov)
    ci, cv = sparse(vm._map)
    vmap = np.asarray(vm._value_map, np.float64)
    return dict(seed=seed, steps=steps, depth_sha256=np.array(hashes), conf_idx=ci, conf_val=cv.astype(np.float32),
                value_val=vmap.reshape(-1)[ci].astype(np.float32), value_sha=np.array(sha(vmap)))


def gen_obstacle_map(ref_om, seed, steps, holes, hole_thresh, kw=None):
    kw = dict(OBSTACLE_KW if kw is None else kw)
    fx, fy, fov = camera_intrinsics(640)
    env = SyntheticEnv(seed, holes=holes)
    om = ref_om.ObstacleMap(hole_area_thresh=hole_thresh, **kw)
    tfs, hashes, fr_px, fr_xy = [], [], [], []
    for _ in range(steps):
        depth, tf, _ = env.observe()
        hashes.append(sha(depth))
        tfs.append(tf)
        om.update_map(depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
        fr_px.append(np.asarray(om._frontiers_px, np.float64).reshape(-1, 2))
        fr_xy.append(np.asarray(om.frontiers, np.float64).reshape(-1, 2))
    cpx, fpx = frontier_blob(fr_px)
    _, fxy = frontier_blob(fr_xy)
    return dict(seed=seed, steps=steps, holes=holes, hole_area_thresh=hole_thresh, fx=fx, fy=fy, fov=fov,
                min_depth=MIN_DEPTH, max_depth=MAX_DEPTH, tf=np.stack(tfs), depth_sha256=np.array(hashes),
                obstacle_bits=packbits(om._map), navigable_bits=packbits(om._navigable_map),
                explored_bits=packbits(om.explored_area), frontier_counts=cpx, frontiers_px=fpx, frontiers_xy=fxy, **kw)


def multicam_script(seed: int, steps: int):
    """Per step: two body cameras (yaw offsets -35 / +35 degrees, max depth 2.5 m like the robot's, reality.yaml:24) that
    only add obstacles, then one reveal from the robot pose without a depth image -- the call pattern of
    RealityMixin._cache_observations (reality_policies.py:113-138).  Shared by the generator and the replaying tests."""
    from vlfm_amd.synthetic import depth_frame, pose_to_tf

    env = SyntheticEnv(seed)
    fx, fy, fov = camera_intrinsics(640)
    for _ in range(steps):
        x, y, yaw = env.traj.step()
        cams = [(depth_frame(env.rng, 480, 640, holes=True), pose_to_tf(x, y, yaw + off)) for off in (-0.61, 0.61)]
        yield cams, pose_to_tf(x, y, yaw), fx, fy, fov


def gen_multicam(ref_om, seed=15, steps=8):
    om = ref_om.ObstacleMap(hole_area_thresh=100000, min_height=0.1, max_height=1.5, agent_radius=0.2, area_thresh=1.5)
    fr_xy, hashes = [], []
    for cams, tf_robot, fx, fy, fov in multicam_script(seed, steps):
        for depth, tf in cams:
            hashes.append(sha(depth))
            om.update_map(depth.copy(), tf, MIN_DEPTH, 2.5, fx, fy, fov, explore=False)
        om.update_map(None, tf_robot, MIN_DEPTH, 2.5, fx, fy, 2 * fov, explore=True, update_obstacles=False)
        fr_xy.append(np.asarray(om.frontiers, np.float64).reshape(-1, 2))
    counts, cat = frontier_blob(fr_xy)
    return dict(seed=seed, steps=steps, depth_sha256=np.array(hashes), frontier_counts=counts, frontiers_xy=cat,
                obstacle_bits=packbits(om._map), navigable_bits=packbits(om._navigable_map),
                explored_bits=packbits(om.explored_area))


def island_script(seed: int = 17):
    """Depth frames whose zero regions ENCLOSE valid texels inside the obstacle height band ("islands" below the image
    centre, where a wall 1.5-3 m away sits 0-0.27 m under the camera): fill_small_holes draws each small contour FILLED
    (img_utils.py:385), so the enclosed valid texels become 1.0 and are dropped as well.  Steps: plain frame; the same
    pose with a small ring (its island cells were already set by the plain frame and must stay); new poses with a small
    ring, a ring whose outer contour is too large to fill but whose inner (hole) contour is not, two nested rings, and a
    ring cut by the image border.  Shared by the generator and the replaying tests."""
    from vlfm_amd.synthetic import depth_frame, pose_to_tf

    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:480, 0:640]

    def ring(d, cx, cy, r_out, r_in):
        rr = (xx - cx) ** 2 + (yy - cy) ** 2
        d[(rr <= r_out ** 2) & (rr > r_in ** 2)] = 0.0

    def wall(z):  # flat wall z metres away + floor, so that rows 240..~290 are inside the 0.61-0.88 m band
        d = depth_frame(rng, 480, 640)
        rows = np.arange(480)[:, None] - 240
        floor = np.where(rows > 0, 0.88 * camera_intrinsics(640)[1] / np.maximum(rows, 1e-9), np.inf)
        return np.clip((np.minimum(z, floor) - MIN_DEPTH) / (MAX_DEPTH - MIN_DEPTH), 1e-3, 1.0).astype(np.float32) \
            + 0 * d

    frames = []
    d0 = wall(2.0)
    frames.append((d0.copy(), pose_to_tf(0.0, 0.0, 0.0)))
    d1 = d0.copy(); ring(d1, 320, 262, 30, 14)
    frames.append((d1, pose_to_tf(0.0, 0.0, 0.0)))
    d2 = wall(2.5); ring(d2, 200, 265, 36, 20); ring(d2, 470, 258, 22, 9)
    frames.append((d2, pose_to_tf(0.3, -0.2, 0.6)))
    d3 = wall(1.8); ring(d3, 320, 250, 215, 150)          # outer area > 100000 px: only the inner contour is filled
    frames.append((d3, pose_to_tf(-0.4, 0.5, -1.1)))
    d4 = wall(2.2); ring(d4, 300, 270, 60, 48); ring(d4, 300, 270, 30, 12)   # nested: island ring holds a second ring
    frames.append((d4, pose_to_tf(1.0, 1.0, 2.4)))
    d5 = wall(2.0); ring(d5, 5, 262, 40, 22); d5[300:330, 100:160] = 0.0     # ring cut by the border + a plain hole
    frames.append((d5, pose_to_tf(-1.0, -1.5, 3.0)))
    frames.append((wall(3.0), pose_to_tf(-1.0, -1.5, 3.0)))                  # a clean frame after island frames
    return frames


def gen_islands(ref_om):
    fx, fy, fov = camera_intrinsics(640)
    om = ref_om.ObstacleMap(hole_area_thresh=100000, **OBSTACLE_KW)
    hashes, fr_xy, per_step = [], [], []
    for depth, tf in island_script():
        hashes.append(sha(depth))
        om.update_map(depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
        fr_xy.append(np.asarray(om.frontiers, np.float64).reshape(-1, 2))
        per_step.append(packbits(om._map))
    counts, cat = frontier_blob(fr_xy)
    return dict(depth_sha256=np.array(hashes), frontier_counts=counts, frontiers_xy=cat,
                obstacle_bits_per_step=np.stack(per_step), navigable_bits=packbits(om._navigable_map),
                explored_bits=packbits(om.explored_area))


EP500_SNAPSHOTS = (100, 250, 500)


def quantize16(a: np.ndarray) -> np.ndarray:
    """[0, 1] floats -> u16 with step 1/65535 (|error| <= 7.7e-6): keeps the 500-step snapshots small; the exact arrays are
    pinned by SHA-256 digests next to them."""
    return np.rint(np.clip(np.asarray(a, np.float64), 0.0, 1.0) * 65535.0).astype(np.uint16)


def gen_episode500(ref_vm, ref_om):
    """One full-length episode (500 steps, pointnav_depth_hm3d.yaml:14) through THE REFERENCE'S ObstacleMap + ValueMap in
    the rooms-and-pillars world of world500.py: 13-22 simultaneous frontiers for most of the run, free-standing pillars
    and non-convex blocks inside the view cone, an explored area that closes around obstacles.  Per step: frontier
    pixels (bit-exact bar) and sort_waypoints values; at steps 100 / 250 / 500: all five planes."""
    import world500 as w5

    fx, fy, fov = camera_intrinsics(w5.W)
    om = ref_om.ObstacleMap(**OBSTACLE_KW)
    vm = ref_vm.ValueMap(1, use_max_confidence=False)
    actions = w5.plan_actions()
    digests, fr_px, sorted_vals, sorted_idx = [], [], [], []
    out = dict(actions=actions, snapshots=np.array(EP500_SNAPSHOTS, np.int32))
    for i, depth, tf, values in w5.episode(actions):
        digests.append(np.frombuffer(hashlib.sha256(depth.tobytes()).digest()[:8], np.uint64)[0])
        om.update_map(depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
        vm.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fov)
        px = np.asarray(om._frontiers_px, np.float64).reshape(-1, 2)
        fr_px.append(px)
        if len(px):
            s_wp, s_val = vm.sort_waypoints(om.frontiers, 0.5)
            # the permutation sort_waypoints applied, recovered by matching rows (frontiers of one step are distinct)
            idx = [int(np.flatnonzero((om.frontiers == p).all(axis=1))[0]) for p in np.asarray(s_wp)]
            sorted_idx.append(np.array(idx, np.int16))
            sorted_vals.append(np.asarray(s_val, np.float64))
        step = i + 1
        if step in EP500_SNAPSHOTS:
            conf = np.asarray(vm._map, np.float32)
            val = np.asarray(vm._value_map, np.float64)[:, :, 0]
            support = conf > 0
            assert not np.any(val[~support])
            out[f"s{step}_support"] = packbits(support)
            out[f"s{step}_conf_q"] = quantize16(conf[support])
            out[f"s{step}_value_q"] = quantize16(val[support])
            out[f"s{step}_conf_sha"] = np.array(sha(conf))
            out[f"s{step}_value_sha"] = np.array(sha(val))
            out[f"s{step}_obstacle"] = packbits(om._map)
            out[f"s{step}_navigable"] = packbits(om._navigable_map)
            out[f"s{step}_explored"] = packbits(om.explored_area)
    counts, cat = frontier_blob(fr_px)
    out.update(depth_digest=np.array(digests, np.uint64), frontier_counts=counts, frontiers_px=cat,
               sorted_idx=np.concatenate(sorted_idx), sorted_values=np.concatenate(sorted_vals),
               value_dtype=str(vm._value_map.dtype), **OBSTACLE_KW)
    return out


def gen_sync(ref_vm, ref_om, seed, steps):
    """reality-style ValueMap(obstacle_map=...) (value_map.py:369-375): full-map zeroing by the explored area."""
    fx, fy, fov = camera_intrinsics(640)
    env = SyntheticEnv(seed)
    om = ref_om.ObstacleMap(**OBSTACLE_KW)
    vm = ref_vm.ValueMap(1, use_max_confidence=False, obstacle_map=om)
    tfs, vals, hashes = [], [], []
    for _ in range(steps):
        depth, tf, values = env.observe()
        hashes.append(sha(depth))
        tfs.append(tf)
        vals.append(values)
        om.update_map(depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
        vm.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fov)
    ci, cv = sparse(vm._map)
    vi, vv = sparse(vm._value_map)
    return dict(seed=seed, steps=steps, fx=fx, fy=fy, fov=fov, min_depth=MIN_DEPTH, max_depth=MAX_DEPTH,
                tf=np.stack(tfs), values=np.stack(vals), depth_sha256=np.array(hashes), conf_idx=ci,
                conf_val=cv.astype(np.float32), value_idx=vi, value_val=np.asarray(vv, np.float64),
                explored_bits=packbits(om.explored_area), **{k: v for k, v in OBSTACLE_KW.items()})


def gen_helpers(geo, img, ref_vm):
    """Small known-answer tables of the in-tree helpers (geometry_utils / img_utils / BaseMap)."""
    rng = np.random.Generator(np.random.PCG64(2024))
    yaws = rng.uniform(-np.pi, np.pi, 8)
    xyz = rng.uniform(-10, 10, (8, 3))
    tfs = np.stack([geo.xyz_yaw_to_tf_matrix(p, y) for p, y in zip(xyz, yaws)])
    ex_yaw = np.array([geo.extract_yaw(t) for t in tfs])
    fovs = np.array([geo.get_fov(388.19, 640), geo.get_fov(500.0, 1280)])
    depth = rng.uniform(0.5, 5.0, (6, 8)).astype(np.float32)
    mask = depth < 4.0
    cloud = geo.get_point_cloud(depth, mask, 7.5, 7.25)
    moved = geo.transform_points(tfs[0], cloud)
    bm = ref_vm.ValueMap(1, size=200)
    pts = rng.uniform(-4.9, 4.9, (16, 2))
    pts[:4] = [[0.025, -0.025], [0.075, 0.125], [-0.025, 0.025], [1.225, -2.375]]  # rint half-way cases
    px = bm._xy_to_px(pts)
    back = bm._px_to_xy(px)
    # pixel_value_within_radius incl. the clipped-crop quirk (img_utils.py:243-253) and the empty (-1) case
    field = np.zeros((60, 60), np.float32)
    field[5:40, 3:33] = rng.uniform(0.1, 1.0, (35, 30)).astype(np.float32)
    cells = np.array([[20, 20], [3, 4], [58, 58], [0, 30], [30, 0], [39, 32]])
    med = np.array([img.pixel_value_within_radius(field, tuple(c), 10) for c in cells], np.float64)
    # rotate_image / place_img_in_img
    tile = np.zeros((21, 21))
    tile[3:15, 8:14] = rng.uniform(0.2, 1.0, (12, 6))
    rot = np.stack([img.rotate_image(tile, a) for a in (0.0, 0.3, -1.1, np.pi / 2)])
    canvas = img.place_img_in_img(np.zeros((30, 30), np.float32), tile, 4, 27)
    return dict(yaws=yaws, xyz=xyz, tfs=tfs, extract_yaw=ex_yaw, fovs=fovs, depth=depth, mask=mask, cloud=cloud,
                moved=moved, pts=pts, px=px, back=back, field=field, cells=cells, medians=med, tile=tile, rotated=rot,
                placed=canvas)


def gen_detections(det_mod):
    """ObjectDetections of the reference (vlfm/vlm/detections.py:15-126): construction, both filters, JSON round trip."""
    import torch

    rng = np.random.Generator(np.random.PCG64(11))
    boxes = torch.from_numpy(rng.uniform(0.05, 0.6, (12, 4)).astype(np.float32))
    logits = torch.from_numpy(rng.uniform(0.1, 0.95, 12).astype(np.float32))
    logits[3] = 0.8  # exactly on the threshold: kept by >=
    names = ["chair", "bed", "potted plant", "tv"]
    phrases = [names[i % 4] for i in range(12)]
    d = det_mod.ObjectDetections(boxes, logits, list(phrases), image_source=None)
    out = dict(in_boxes=boxes.numpy(), in_logits=logits.numpy(), in_phrase_idx=np.arange(12) % 4,
               boxes_xyxy=d.boxes.numpy().copy())
    d.filter_by_class(["chair", "tv", "sofa"])
    out.update(after_class_boxes=d.boxes.numpy().copy(), after_class_phrases=np.array(d.phrases))
    d.filter_by_conf(0.8)
    j = d.to_json()
    out.update(after_conf_boxes=np.array(j["boxes"], np.float64).reshape(-1, 4), after_conf_logits=np.array(j["logits"]),
               after_conf_phrases=np.array(j["phrases"]), num=d.num_detections)
    r = det_mod.ObjectDetections.from_json(j)
    out.update(roundtrip_boxes=r.boxes.numpy().copy())
    return out


def object_map_script():
    """Seeded sequence of ObjectPointCloudMap calls shared by the generator and the replaying tests: (op, args) tuples."""
    rng = np.random.Generator(np.random.PCG64(21))
    from vlfm_amd.synthetic import depth_frame, pose_to_tf

    H, W = 480, 640
    yy, xx = np.mgrid[0:H, 0:W]
    ops = []
    blobs = [(320, 250, 60, 45, 0.0), (150, 300, 50, 70, 0.3), (30, 240, 25, 60, 0.6), (600, 260, 30, 50, 0.9),
             (330, 260, 120, 90, 1.2), (320, 250, 8, 8, 1.5)]
    for k, (cx, cy, ax, ay, yaw) in enumerate(blobs):
        depth = depth_frame(rng, H, W)
        depth[(xx - cx) ** 2 / (1.3 * ax) ** 2 + (yy - cy) ** 2 / (1.3 * ay) ** 2 <= 1] = 0.35 + 0.08 * k  # an object in front of the wall
        if k == 1:
            depth[cy - 5:cy + 5, cx - 5:cx + 5] = 0.0  # holes inside the mask -> "far"
        mask = ((xx - cx) ** 2 / ax ** 2 + (yy - cy) ** 2 / ay ** 2 <= 1).astype(np.uint8)
        tf = pose_to_tf(0.2 * k, -0.1 * k, yaw)
        ops.append(("update", "chair" if k % 2 == 0 else "bed", depth, mask, tf))
        ops.append(("best", "chair", np.array([0.2 * k, -0.1 * k])))
        if k == 3:
            ops.append(("explored", pose_to_tf(1.0, 0.5, 0.4)))
    return ops


def gen_object_map(om_mod):
    """ObjectPointCloudMap of the reference (object_point_cloud_map.py) driven by object_map_script() with NumPy's global
    RNG seeded: cloud per class after every operation, get_best_object results."""
    fx, fy, fov = camera_intrinsics(640)
    np.random.seed(1234)
    m = om_mod.ObjectPointCloudMap(erosion_size=5)
    out = {}
    for i, op in enumerate(object_map_script()):
        if op[0] == "update":
            m.update_map(op[1], op[2], op[3], op[4], MIN_DEPTH, MAX_DEPTH, fx, fy)
        elif op[0] == "best":
            if m.has_object(op[1]):
                out[f"best_{i}"] = np.asarray(m.get_best_object(op[1], op[2]), np.float64)
        else:
            m.update_explored(op[1], MAX_DEPTH, fov)
        for name in ("chair", "bed"):   # per operation: a digest; the full clouds only once, at the end
            if name in m.clouds:
                c = np.asarray(m.clouds[name], np.float64)
                out[f"sig_{i}_{name}"] = np.array([str(c.shape[0]), sha(c)])
    for name in ("chair", "bed"):
        out[f"final_{name}"] = np.asarray(m.clouds[name], np.float64)
    return out


OBJECT_MAP_RANDOM_SEEDS = (0, 1, 2)


def object_map_random_script(seed: int):
    """A RANDOM session for ObjectPointCloudMap (round 6): three classes, blobs of random size and place -- some thinner than the
    erosion (empty cloud), some at the far plane (the "too far" tagging draws from NumPy's global RNG), some cut by the image border,
    depth holes inside masks -- random poses, `update_explored` and `get_best_object` calls in between."""
    rng = np.random.Generator(np.random.PCG64(900 + seed))
    from vlfm_amd.synthetic import depth_frame, pose_to_tf

    H, W = 480, 640
    yy, xx = np.mgrid[0:H, 0:W]
    names = ("chair", "bed", "tv")
    ops = []
    x = y = 0.0
    for k in range(10):
        x += float(rng.uniform(-0.4, 0.4))
        y += float(rng.uniform(-0.4, 0.4))
        yaw = float(rng.uniform(-np.pi, np.pi))
        depth = depth_frame(rng, H, W, holes=bool(rng.integers(0, 2)))
        cx, cy = int(rng.integers(-20, W + 20)), int(rng.integers(-20, H + 20))
        ax, ay = (int(rng.integers(2, 9)), int(rng.integers(2, 9))) if k % 4 == 3 else (int(rng.integers(15, 140)), int(rng.integers(15, 110)))
        blob = (xx - cx) ** 2 / ax ** 2 + (yy - cy) ** 2 / ay ** 2 <= 1
        near = float(rng.uniform(0.05, 0.95)) if k % 5 else 1.0          # every fifth object sits at the far plane
        depth[(xx - cx) ** 2 / (1.3 * ax) ** 2 + (yy - cy) ** 2 / (1.3 * ay) ** 2 <= 1] = near
        if k % 3 == 1 and blob.any():
            depth[max(cy - 4, 0):cy + 4, max(cx - 4, 0):cx + 4] = 0.0     # a hole inside the mask -> "far"
        mask = blob.astype(np.uint8)
        tf = pose_to_tf(x, y, yaw)
        ops.append(("update", names[int(rng.integers(0, 3))], depth, mask, tf))
        ops.append(("best", names[int(rng.integers(0, 3))], np.array([x, y])))
        if k % 3 == 2:
            ops.append(("explored", pose_to_tf(x + float(rng.uniform(-1, 1)), y + float(rng.uniform(-1, 1)), float(rng.uniform(-np.pi, np.pi)))))
    return ops


def gen_object_map_random(om_mod, seed: int):
    """The reference's ObjectPointCloudMap over object_map_random_script(seed): a digest of every class's cloud after every operation,
    every get_best_object result, has_object per class, the final clouds."""
    fx, fy, fov = camera_intrinsics(640)
    np.random.seed(4321 + seed)
    m = om_mod.ObjectPointCloudMap(erosion_size=int((3, 5, 2)[seed % 3]))
    m.reset()       # `clouds` is a CLASS attribute of the reference (object_point_cloud_map.py:18): a new instance sees the previous one's
    out = {}
    for i, op in enumerate(object_map_random_script(seed)):
        if op[0] == "update":
            m.update_map(op[1], op[2], op[3], op[4], MIN_DEPTH, MAX_DEPTH, fx, fy)
        elif op[0] == "best":
            out[f"has_{i}"] = np.array([int(m.has_object(n)) for n in ("chair", "bed", "tv")])
            if m.has_object(op[1]):
                out[f"best_{i}"] = np.asarray(m.get_best_object(op[1], op[2]), np.float64)
        else:
            m.update_explored(op[1], MAX_DEPTH, fov)
        for name in ("chair", "bed", "tv"):
            if name in m.clouds:
                c = np.asarray(m.clouds[name], np.float64)
                out[f"sig_{i}_{name}"] = np.array([str(c.shape[0]), sha(c)])
    for name in ("chair", "bed", "tv"):
        if name in m.clouds:
            out[f"final_{name}"] = np.asarray(m.clouds[name], np.float64)
    return out


def gen_policy(name):
    """One scripted episode through THE REFERENCE'S ``ITMPolicyV2`` (vlfm/policy/itm_policy.py:236-267 on top of
    BaseITMPolicy :26-234 and BaseObjectNavPolicy base_objectnav_policy.py:35-365, real source via
    ref_shim.reference_policy(), over the reference's real ObstacleMap / ValueMap / ObjectPointCloudMap / ObjectDetections).
    Only HabitatMixin (habitat_policies.py, needs habitat) is restated, in the subclass below, with line citations; the
    four model clients are the scripted stand-ins of tests/golden/policy_script.py."""
    import torch

    from oracle import ref_shim

    import policy_script as ps

    itm_mod, base_mod = ref_shim.reference_policy()
    det_mod = ref_shim.reference_detections()
    geo = ref_shim.reference_modules()[2]

    class _CpuTorch:  # BaseObjectNavPolicy._pointnav asks for device="cuda" (base_objectnav_policy.py:254,262): none here
        def __getattr__(self, n):
            return getattr(torch, n)

        @staticmethod
        def tensor(*a, device=None, **k):
            return torch.tensor(*a, **k)

    base_mod.torch = _CpuTorch()
    seed, steps, dataset, goal_name, caption = ps.EPISODES[name]
    vlm = ps.ScriptedVLM(name, det_mod.ObjectDetections)
    log = dict(mode=[], nav_goal=[], best_value=[], last_frontier=[], stop=[], rho_theta=[], n_det=[], mask_px=[],
               frontier_counts=[], frontiers=[], goal_xy=[], pointnav_resets=[])
    TURN_LEFT, STOP = torch.tensor([[2]]), torch.tensor([[0]])  # TorchActionIDs habitat_policies.py:53-57

    class ScriptedHabitatITMPolicyV2(itm_mod.ITMPolicyV2):
        _stop_action = STOP

        def __init__(self, camera_height, min_depth, max_depth, camera_fov, image_width, **kw):
            super().__init__(**kw)
            self._camera_height, self._min_depth, self._max_depth = camera_height, min_depth, max_depth  # :84-91
            self._camera_fov = np.deg2rad(camera_fov)
            self._fx = self._fy = image_width / (2 * np.tan(self._camera_fov / 2))
            self._itm, self._coco_object_detector = vlm.itm, vlm.coco
            self._object_detector, self._mobile_sam = vlm.gdino, vlm.sam

            class _Act:
                resets = 0

                def reset(self):
                    _Act.resets += 1

                def act(self, obs, masks, deterministic=True):
                    assert obs["depth"].shape[1:3] == (224, 224)
                    return torch.tensor([[1]])

            self._pointnav_policy = _Act()

        def _initialize(self):  # habitat_policies.py:150-153
            log["mode"].append("initialize")
            self._done_initializing = not self._num_steps < 11
            return TURN_LEFT

        def _explore(self, observations):
            log["mode"].append("explore")
            return super()._explore(observations)

        def _pointnav(self, goal, stop=False):
            if stop:
                log["mode"].append("navigate")
            return super()._pointnav(goal, stop=stop)

        def _cache_observations(self, observations):  # habitat_policies.py:173-237 (filter_depth: input is pre-filtered)
            if len(self._observations_cache) > 0:
                return
            rgb, depth = observations["rgb"], observations["depth"]
            x, y = observations["gps"]
            camera_yaw = observations["compass"]
            camera_position = np.array([x, -y, self._camera_height])
            robot_xy = camera_position[:2]
            tf = geo.xyz_yaw_to_tf_matrix(camera_position, camera_yaw)
            self._obstacle_map.update_map(depth, tf, self._min_depth, self._max_depth, self._fx, self._fy, self._camera_fov)
            frontiers = self._obstacle_map.frontiers
            self._obstacle_map.update_agent_traj(robot_xy, camera_yaw)
            self._observations_cache = {
                "frontier_sensor": frontiers, "nav_depth": torch.from_numpy(depth.reshape(1, *depth.shape, 1)),
                "robot_xy": robot_xy, "robot_heading": camera_yaw,
                "object_map_rgbd": [(rgb, depth, tf, self._min_depth, self._max_depth, self._fx, self._fy)],
                "value_map_rgbd": [(rgb, depth, tf, self._min_depth, self._max_depth, self._camera_fov)],
                "habitat_start_yaw": camera_yaw,
            }

    cfg = base_mod.VLFMConfig()
    kw = {k: getattr(cfg, k) for k in base_mod.VLFMConfig.kwaarg_names}  # the reference's own defaults
    kw.update(visualize=False)
    np.random.seed(777)  # ObjectPointCloudMap subsamples / tags with NumPy's global RNG
    pol = ScriptedHabitatITMPolicyV2(camera_height=0.88, min_depth=MIN_DEPTH, max_depth=MAX_DEPTH, camera_fov=79.0,
                                     image_width=ps.W, **kw)
    if dataset == "mp3d":
        pol._non_coco_caption = caption  # HabitatMixin.act habitat_policies.py:139-141
    import contextlib
    import io

    world = ps.ScriptedWorld(name)
    for _ in range(steps):
        k, rgb, depth, x, y, yaw = world.observe()
        obs = {"rgb": rgb, "depth": depth, "gps": (x, -y), "compass": yaw, "objectgoal": goal_name}
        masks = torch.tensor([[k != 0]])
        with contextlib.redirect_stdout(io.StringIO()):
            try:
                pol.act(obs, None, None, masks)
            except StopIteration:  # habitat_policies.py:144-145
                log["mode"].append("edge_of_map")
        fr = np.asarray(pol._obstacle_map.frontiers, np.float64).reshape(-1, 2)
        log["frontier_counts"].append(len(fr))
        log["frontiers"].append(fr)
        log["nav_goal"].append(np.asarray(pol._last_goal, np.float64))
        log["best_value"].append(float(pol._last_value))
        log["last_frontier"].append(np.asarray(pol._last_frontier, np.float64))
        log["stop"].append(bool(pol._called_stop))
        log["rho_theta"].append(np.asarray(pol._policy_info.get("rho_theta", [np.nan, np.nan]), np.float64))
        world.advance(log["mode"][-1], *log["rho_theta"][-1])
        log["mask_px"].append(int(pol._object_masks.sum()))
        log["pointnav_resets"].append(int(pol._pointnav_policy.resets))
        has = pol._object_map.has_object(goal_name)
        log["goal_xy"].append(np.asarray(pol._object_map.last_target_coord if has and pol._object_map.last_target_coord
                                         is not None else [np.nan, np.nan], np.float64))
    conf_idx, conf_val = sparse(pol._value_map._map)
    vmap = np.asarray(pol._value_map._value_map, np.float64)
    assert not np.any(vmap.reshape(-1)[np.setdiff1d(np.arange(vmap.size), conf_idx)])  # value support within conf support
    cloud = pol._object_map.clouds.get(goal_name, np.zeros((0, 4)))
    return dict(
        pose=np.array(world.poses, np.float64), wall=np.array(world.walls, np.float32),
        mode=np.array(log["mode"]), nav_goal=np.array(log["nav_goal"]), best_value=np.array(log["best_value"]),
        last_frontier=np.array(log["last_frontier"]), stop=np.array(log["stop"]), rho_theta=np.array(log["rho_theta"]),
        mask_px=np.array(log["mask_px"], np.int64), pointnav_resets=np.array(log["pointnav_resets"], np.int32),
        frontier_counts=np.array(log["frontier_counts"], np.int32),
        frontiers=np.concatenate(log["frontiers"]) if sum(log["frontier_counts"]) else np.zeros((0, 2)),
        goal_xy=np.array(log["goal_xy"]), prompts=np.array(vlm.prompts), captions=np.array(vlm.captions or [""]),
        calls=np.array([f"{k}:{w}" for k, w in vlm.calls]),
        conf_idx=conf_idx, conf_val=conf_val, value_val=vmap.reshape(-1)[conf_idx].astype(np.float32),  # f32: 1e-4 bar
        value_sha=np.array(sha(vmap)),  # ... and the exact f64 map as a digest for the bit-exact CPU comparison
        cloud_sig=np.array([str(len(cloud)), sha(np.asarray(cloud, np.float64))]),
        explored=packbits(pol._obstacle_map.explored_area), obstacles=packbits(pol._obstacle_map._map))


API = {
    # reference module -> {class: [methods]} -- the drop-in boundary of SURVEY.md section 8(b)
    "vlfm.mapping.base_map": {"BaseMap": ["__init__", "reset", "update_agent_traj", "_xy_to_px", "_px_to_xy"]},
    "vlfm.mapping.value_map": {"ValueMap": ["__init__", "reset", "update_map", "sort_waypoints", "visualize"]},
    "vlfm.mapping.obstacle_map": {"ObstacleMap": ["__init__", "reset", "update_map", "visualize"]},
    "vlfm.mapping.object_point_cloud_map": {"ObjectPointCloudMap": [
        "__init__", "reset", "has_object", "update_map", "get_best_object", "update_explored", "get_target_cloud"]},
    "vlfm.vlm.blip2itm": {"BLIP2ITM": ["__init__", "cosine"], "BLIP2ITMClient": ["__init__", "cosine"]},
    "vlfm.vlm.yolov7": {"YOLOv7": ["__init__", "predict"], "YOLOv7Client": ["__init__", "predict"]},
    "vlfm.vlm.grounding_dino": {"GroundingDINO": ["__init__", "predict"], "GroundingDINOClient": ["__init__", "predict"]},
    "vlfm.vlm.sam": {"MobileSAM": ["__init__", "segment_bbox"], "MobileSAMClient": ["__init__", "segment_bbox"]},
    "vlfm.vlm.detections": {"ObjectDetections": ["__init__", "filter_by_conf", "filter_by_class", "to_json", "from_json"]},
}


def signature_table(resolve):
    """{"module.Class.method": [[name, kind, default-repr or None], ...]} via inspect; ``resolve(module)`` imports it."""
    import inspect

    table = {}
    for mod_name, classes in API.items():
        mod = resolve(mod_name)
        for cls_name, methods in classes.items():
            cls = getattr(mod, cls_name)
            for m in methods:
                sig = inspect.signature(getattr(cls, m))
                table[f"{mod_name}.{cls_name}.{m}"] = [
                    [p.name, p.kind.name, None if p.default is inspect.Parameter.empty else
                     ("<function>" if inspect.isfunction(p.default) else repr(p.default))]  # no addresses in fixtures
                    for p in sig.parameters.values()]
    return table


def gen_api():
    import importlib
    import json

    from oracle import ref_shim

    ref_shim.reference_policy()  # installs every stand-in the vlm modules need at import time
    return {"json": np.array(json.dumps(signature_table(importlib.import_module), sort_keys=True))}


def gen_pointnav():
    """THE REFERENCE'S PointNav network (vlfm/policy/utils/non_habitat_policy/nh_pointnav_policy.py -- pure PyTorch, runs
    here unmodified) and its depth transform (vlfm/obs_transformers/utils.py:image_resize), one environment at a time as
    the reference's wrapper does (pointnav_policy.py:50-128), on the pattern weights / inputs of pointnav_script.py."""
    import importlib

    import torch

    from oracle import ref_shim

    import pointnav_script as pn

    ref_shim.install()
    nh = importlib.import_module("vlfm.policy.utils.non_habitat_policy.nh_pointnav_policy")
    resize = importlib.import_module("vlfm.obs_transformers.utils").image_resize
    torch.manual_seed(0)
    policy = nh.PointNavResNetPolicy().eval()
    with torch.no_grad():
        pn.pattern_(policy.state_dict())
        state = [torch.zeros(1, 4, 512) for _ in range(pn.N_ENVS)]
        prev = [torch.zeros(1, 2) for _ in range(pn.N_ENVS)]
        actions = []
        for depth, rt, masks in pn.inputs():
            row = []
            for e in range(pn.N_ENVS):
                obs = {"depth": resize(depth[e:e + 1].unsqueeze(-1), (224, 224), channels_last=True,
                                       interpolation_mode="area"),
                       "pointgoal_with_gps_compass": rt[e:e + 1]}
                a, state[e] = policy.act(obs, state[e], prev[e], masks[e].view(1, 1), deterministic=True)
                prev[e] = a.clone()
                row.append(a[0].numpy().copy())
            actions.append(np.stack(row))
    keys = sorted(policy.state_dict().keys())
    return dict(actions=np.stack(actions).astype(np.float32), final_state=torch.cat(state).numpy().astype(np.float32),
                state_dict_keys=np.array(keys),
                state_dict_shapes=np.array([str(tuple(policy.state_dict()[k].shape)) for k in keys]))


def _ep500_worker():
    from oracle import ref_shim

    ref_vm, ref_om, _, _ = ref_shim.reference_modules()
    return gen_episode500(ref_vm, ref_om)


EPISODE_LOG_SAMPLE = {"failure_cause": "did_not_fail", "success": 1, "spl": 0.73, "distance_to_goal": 0.4,
                      "num_steps": 212, "target_object": "potted plant", "traveled_stairs": False,
                      "nested": {"rho_theta": [1.5, -0.25], "stop_called": True}}


def gen_episode_log():
    """THE REFERENCE'S log_saver (vlfm/utils/log_saver.py -- standard library only, runs here as it is): the bytes and the
    file name of one episode log, the skip-if-present rule and is_evaluated."""
    import importlib.util
    import tempfile

    spec = importlib.util.spec_from_file_location("ref_log_saver", "/root/reference/vlfm/utils/log_saver.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import contextlib
    import io

    with tempfile.TemporaryDirectory() as d:
        old = os.environ.get("ZSOS_LOG_DIR")
        os.environ["ZSOS_LOG_DIR"] = os.path.join(d, "logs")
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                before = mod.is_evaluated(17, "scene_a")
                mod.log_episode(17, "scene_a", EPISODE_LOG_SAMPLE)
                mod.log_episode(17, "scene_a", {"failure_cause": "overwritten?"})   # must be skipped
                after = mod.is_evaluated(17, "scene_a")
            names = sorted(os.listdir(os.environ["ZSOS_LOG_DIR"]))
            text = open(os.path.join(os.environ["ZSOS_LOG_DIR"], names[0])).read()
        finally:
            if old is None:
                del os.environ["ZSOS_LOG_DIR"]
            else:
                os.environ["ZSOS_LOG_DIR"] = old
    return dict(file_names=np.array(names), text=np.array(text), evaluated_before=before, evaluated_after=after)


def generate():
    from concurrent.futures import ProcessPoolExecutor
    import multiprocessing

    from oracle import ref_shim

    # the 500-step episode takes about a minute of the reference's full-map NumPy passes: its own process, beside the rest
    pool = ProcessPoolExecutor(1, mp_context=multiprocessing.get_context("spawn"))
    ep500 = pool.submit(_ep500_worker)
    ref_vm, ref_om, geo, img = ref_shim.reference_modules()
    out = {}
    for name, args in VM_CASES.items():
        out[name] = gen_value_map(ref_vm, *args)
    for name, args in OM_CASES.items():
        out[name] = gen_obstacle_map(ref_om, *args)
    out[SYNC_CASE[0]] = gen_sync(ref_vm, ref_om, SYNC_CASE[1], SYNC_CASE[2])
    out["om_multicam"] = gen_multicam(ref_om)
    out["om_islands"] = gen_islands(ref_om)
    out["vm_two_cameras"] = gen_two_cameras(ref_vm)
    out["helpers"] = gen_helpers(geo, img, ref_vm)
    out["detections"] = gen_detections(ref_shim.reference_detections())
    out["object_map"] = gen_object_map(ref_shim.reference_object_map())
    for seed in OBJECT_MAP_RANDOM_SEEDS:
        out[f"object_map_rand{seed}"] = gen_object_map_random(ref_shim.reference_object_map(), seed)
    import policy_script as ps

    for name in ps.EPISODES:
        out[name] = gen_policy(name)
    out["api_signatures"] = gen_api()
    out["pointnav"] = gen_pointnav()
    out["episode_log"] = gen_episode_log()
    out["ep500"] = ep500.result()
    pool.shutdown()
    return out


def same(a, b) -> bool:
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind in "US" or b.dtype.kind in "US":
        return a.shape == b.shape and bool(np.all(a.astype(str) == b.astype(str)))
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


def describe_difference(have, want) -> str:
    """Largest difference and first differing cell of two arrays (for the --check report)."""
    a, b = np.asarray(have), np.asarray(want)
    if a.shape != b.shape:
        return f"shape {a.shape} in the file, {b.shape} regenerated"
    if a.dtype.kind in "US" or b.dtype.kind in "US" or a.dtype.kind == "b":
        bad = np.flatnonzero((a != b).reshape(-1))
        return f"{len(bad)} of {a.size} entries differ, first at flat index {int(bad[0])}: {a.reshape(-1)[bad[0]]!r} vs {b.reshape(-1)[bad[0]]!r}"
    neq = ~((a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64))))
    bad = np.flatnonzero(neq.reshape(-1))
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).reshape(-1)
    first = np.unravel_index(int(bad[0]), a.shape) if a.ndim else ()
    return (f"{len(bad)} of {a.size} cells differ, max |diff| {np.nanmax(d[bad]):.3e}, first at {tuple(int(i) for i in first)}: "
            f"{a.reshape(-1)[bad[0]]!r} in the file vs {b.reshape(-1)[bad[0]]!r} regenerated")


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--real-cv2", action="store_true",
                    help="run the reference on the real cv2 (mandatory) / frontier_exploration (if importable) instead of the stand-ins")
    args = ap.parse_args()
    if args.real_cv2:
        if not args.check:
            ap.error("--real-cv2 only makes sense with --check: the committed fixtures are the stand-in run everyone can reproduce")
        os.environ["VLFM_REAL_CV2"] = "1"      # before oracle.cv is imported, and inherited by the ep500 worker process
    from oracle import ref_shim

    cases = generate()
    print("libraries under the reference:", ref_shim.backends())
    bad = 0
    for name, blob in cases.items():
        path = os.path.join(HERE, name + ".npz")
        if args.check:
            here = 0
            with np.load(path, allow_pickle=False) as have:
                for k, v in blob.items():
                    if k not in have.files:
                        print(f"MISMATCH {name}.{k}: not in the committed file")
                        here += 1
                    elif not same(have[k], v):
                        print(f"MISMATCH {name}.{k}: {describe_difference(have[k], v)}")
                        here += 1
            bad += here
            print(f"checked {name}: {'ok' if not here else 'DIFFERS'}")
        else:
            np.savez_compressed(path, **blob)
            print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

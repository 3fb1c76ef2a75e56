"""-m gpu: ObjectPointCloudMap (HIP erosion + masked back-projection + DBSCAN) against the fixture produced by the
reference's own object_point_cloud_map.py (tests/golden/make_golden.py: cv2.erode/boundingRect and open3d stand-ins), and
the device kernels against their oracle restatements.  Clouds must match BIT-EXACTLY (f64) including order."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch

from golden_util import load
from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, camera_intrinsics

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_object_map_replays_reference_fixture(gpu_device):
    from make_golden import object_map_script
    from vlfm_amd.mapping.object_point_cloud_map import ObjectPointCloudMap

    g = load("object_map")
    fx, fy, fov = camera_intrinsics(640)
    np.random.seed(1234)   # the reference draws its subsample / range ids from NumPy's global RNG; so do we
    m = ObjectPointCloudMap(erosion_size=5, device=gpu_device)
    for i, op in enumerate(object_map_script()):
        if op[0] == "update":
            m.update_map(op[1], op[2], op[3], op[4], MIN_DEPTH, MAX_DEPTH, fx, fy)
        elif op[0] == "best":
            if m.has_object(op[1]):
                assert np.array_equal(np.asarray(m.get_best_object(op[1], op[2]), np.float64), g[f"best_{i}"]), i
        else:
            m.update_explored(op[1], MAX_DEPTH, fov)
        for name in ("chair", "bed"):
            if name in m.clouds:
                c = np.asarray(m.clouds[name], np.float64)
                assert [str(c.shape[0]), _sha(c)] == list(g[f"sig_{i}_{name}"]), (i, name, c.shape)
    for name in ("chair", "bed"):
        assert np.array_equal(np.asarray(m.clouds[name], np.float64), g[f"final_{name}"])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_object_map_replays_random_reference_sessions(gpu_device, seed):
    """Round 6: three RANDOM sessions (tests/golden/make_golden.py:object_map_random_script -- three classes, blobs thinner than the
    erosion, objects at the far plane whose points get the random "too far" tag, masks cut by the image border, depth holes, random
    poses, update_explored in between) recorded from THE REFERENCE'S ObjectPointCloudMap: every cloud after every operation (row count +
    SHA-256 of the f64 array), has_object per class, every get_best_object result, the final clouds."""
    from make_golden import object_map_random_script
    from vlfm_amd.mapping.object_point_cloud_map import ObjectPointCloudMap

    g = load(f"object_map_rand{seed}")
    fx, fy, fov = camera_intrinsics(640)
    np.random.seed(4321 + seed)
    m = ObjectPointCloudMap(erosion_size=int((3, 5, 2)[seed % 3]), device=gpu_device)
    m.reset()
    n_best = 0
    for i, op in enumerate(object_map_random_script(seed)):
        if op[0] == "update":
            m.update_map(op[1], op[2], op[3], op[4], MIN_DEPTH, MAX_DEPTH, fx, fy)
        elif op[0] == "best":
            assert [int(m.has_object(n)) for n in ("chair", "bed", "tv")] == list(g[f"has_{i}"]), i
            if m.has_object(op[1]):
                assert np.array_equal(np.asarray(m.get_best_object(op[1], op[2]), np.float64), g[f"best_{i}"]), i
                n_best += 1
        else:
            m.update_explored(op[1], MAX_DEPTH, fov)
        for name in ("chair", "bed", "tv"):
            assert (name in m.clouds) == (f"sig_{i}_{name}" in g), (i, name)
            if name in m.clouds:
                c = np.asarray(m.clouds[name], np.float64)
                assert [str(c.shape[0]), _sha(c)] == list(g[f"sig_{i}_{name}"]), (i, name, c.shape)
    assert n_best >= 5
    for name in ("chair", "bed", "tv"):
        if f"final_{name}" in g:
            assert np.array_equal(np.asarray(m.clouds[name], np.float64), g[f"final_{name}"])


@pytest.mark.parametrize("n,seed", [(300, 0), (1500, 1), (5000, 2), (64, 3), (129, 4)])
def test_dbscan_matches_sequential_oracle(gpu_device, n, seed):
    """Clustered + noisy point sets: labels up to renaming are checked through the selected largest cluster, which must be
    the same index set in the same order as the sequential Open3D-style algorithm picks."""
    import ctypes

    from oracle.ref_object_map import cluster_dbscan
    from vlfm_amd import _lib

    rng = np.random.default_rng(seed)
    k = max(1, n // 400)
    centres = rng.uniform(-2, 2, (k + 1, 3))
    pts = np.concatenate([centres[i % (k + 1)] + rng.normal(0, 0.12, 3) for i in range(n)]).reshape(n, 3)
    pts[rng.random(n) < 0.1] = rng.uniform(-3, 3, (int((rng.random(n) < 0.1).sum()) or 1, 3))[:1]  # a few stragglers
    for eps, mp in ((0.2, 100), (0.15, 20), (0.3, 5)):
        labels = cluster_dbscan(pts, eps, mp)
        uniq, cnt = np.unique(labels[labels >= 0], return_counts=True)
        want = np.where(labels == uniq[np.argmax(cnt)])[0] if len(uniq) else np.zeros(0, np.int64)
        d = torch.from_numpy(pts).to(gpu_device)
        L = _lib.lib()
        sc = torch.empty(L.vlfm_dbscan_scratch_bytes(n), dtype=torch.uint8, device=gpu_device)
        lab = torch.empty(n, dtype=torch.int32, device=gpu_device)
        keep = torch.empty(n, dtype=torch.int32, device=gpu_device)
        num = torch.zeros(1, dtype=torch.int32, device=gpu_device)
        _lib.check(L.vlfm_dbscan_largest_cluster(d.data_ptr(), n, eps, mp, sc.data_ptr(), sc.numel(), lab.data_ptr(),
                                                 keep.data_ptr(), num.data_ptr(), torch.cuda.current_stream().cuda_stream))
        got = keep[: int(num.item())].cpu().numpy()
        assert np.array_equal(got, want), (n, eps, mp, len(got), len(want))
        # full labelling: same partition (noise <-> INT32_MAX; clusters in the same order)
        gl = lab.cpu().numpy().astype(np.int64)
        roots = np.unique(gl[gl != 0x7FFFFFFF])
        remap = {r: i for i, r in enumerate(roots)}
        mapped = np.array([remap.get(x, -1) for x in gl])
        assert np.array_equal(mapped, labels), (n, eps, mp)


def test_erosion_and_cloud_order(gpu_device):
    from oracle.ref_object_map import extract_object_cloud
    from vlfm_amd.mapping.object_point_cloud_map import ObjectPointCloudMap
    from vlfm_amd.synthetic import depth_frame

    rng = np.random.default_rng(9)
    fx, fy, _ = camera_intrinsics(640)
    depth = depth_frame(rng, 480, 640, holes=True)
    for it, mask in ((0, None), (1, None), (5, None), (3, "edge")):
        yy, xx = np.mgrid[0:480, 0:640]
        m = ((xx - 300) ** 2 / 90 ** 2 + (yy - 200) ** 2 / 60 ** 2 <= 1).astype(np.uint8)
        if mask == "edge":
            m[:] = 0
            m[0:40, 0:70] = 1       # touches the image border: the border must not erode
            m[440:480, 600:640] = 1
            m[100:103, 100:300] = 1  # thinner than the erosion: disappears
        om = ObjectPointCloudMap(erosion_size=it, device=gpu_device)
        om.use_dbscan = False
        np.random.seed(5)
        got = om._extract_object_cloud(depth, m, MIN_DEPTH, MAX_DEPTH, fx, fy)
        np.random.seed(5)
        want = extract_object_cloud(depth, m, it, MIN_DEPTH, MAX_DEPTH, fx, fy, use_dbscan=False)
        assert got.shape == want.shape and np.array_equal(got, want), (it, mask, got.shape, want.shape)
    # a mask the erosion removes entirely, DBSCAN on: the reference's open3d_dbscan_filtering returns np.array([]) (shape (0,)) for an
    # empty cloud as it does for an all-noise one (object_point_cloud_map.py:200-201) -- found by the random runs of round 6
    m = np.zeros((480, 640), np.uint8)
    m[100:103, 100:300] = 1
    for use_db in (True, False):
        om = ObjectPointCloudMap(erosion_size=3, device=gpu_device)
        om.use_dbscan = use_db
        got = om._extract_object_cloud(depth, m, MIN_DEPTH, MAX_DEPTH, fx, fy)
        want = extract_object_cloud(depth, m, 3, MIN_DEPTH, MAX_DEPTH, fx, fy, use_dbscan=use_db)
        assert got.shape == want.shape == ((0,) if use_db else (0, 3))

"""Pins for oracle/ref_value_map.py derived by hand from the reference source.
The reference itself holds no golden vectors for this path; parity with real OpenCV is unpinned."""
import numpy as np
import pytest

from oracle.ref_geometry import pose_to_tf
from oracle.ref_value_map import RefBaseMap, RefValueMap

FOV = np.deg2rad(79)


def test_confidence_profile():
    v = RefValueMap(1)
    m = v._get_confidence_mask(FOV, 5.0)
    assert m.shape == (201, 201) and m.dtype == np.float64
    assert m[150, 100] == 1.0 and m[200, 100] == 1.0          # optical axis
    assert m[99].sum() == 0 and m[:, :30].sum() == 0            # nothing behind / far to the side
    # formula of value_map.py:345-350 at hand-picked cells: conf = 0.75*cos^2(theta*pi/fov) + 0.25
    for (r, c) in [(150, 110), (180, 130), (120, 95), (170, 140), (160, 140)]:
        theta = np.arctan2(abs(c - 100), abs(r - 100))
        want = np.float32(np.cos(theta * (np.pi / 2) / (FOV / 2)) ** 2 * 0.75 + 0.25)
        assert m[r, c] == pytest.approx(float(want), abs=1e-7)
    # at theta = fov/2 the confidence bottoms out at min_confidence = 0.25
    edge = m[m > 0].min()
    assert 0.25 <= edge < 0.26
    # cached copy is independent
    m2 = v._get_confidence_mask(FOV, 5.0)
    m2[:] = 0
    assert v._get_confidence_mask(FOV, 5.0)[150, 100] == 1.0


def test_depth_cut_extremes():
    v = RefValueMap(1)
    full = v._get_confidence_mask(FOV, 5.0)
    # depth == 1 -> profile on the far arc: only the Bresenham outline of the cut polygon (row ~200) is removed
    vis1 = v._process_local_data(np.ones((480, 640), np.float32), FOV, 0.5, 5.0)
    assert np.array_equal(vis1[:200] > 0, full[:200] > 0)
    # depth == 0 -> depth_row = min_depth: profile at row int(0.5*20 + 100.5) = 110, everything from there on is cut
    vis0 = v._process_local_data(np.zeros((480, 640), np.float32), FOV, 0.5, 5.0)
    assert np.array_equal(vis0[:110], full[:110]) and vis0[110:].sum() == 0 and vis0[105:110].sum() > 0
    c = v.depth_profile_contour(np.zeros((480, 640), np.float32), FOV, 0.5, 5.0, (201, 201))
    assert c[0].tolist() == [0, 200] and c[-1].tolist() == [200, 200] and (c[1:-1, 1] == 110).all()
    assert c[1, 0] == int(0.5 * np.tan(-FOV / 2) * 20 + 100.5) and len(c) == 642


def test_fusion_algebra_weighted():
    v = RefValueMap(1, use_max_confidence=False)
    depth = np.ones((480, 640), np.float32)
    tf = pose_to_tf((0, 0, 0.88), 0.0)
    v.update_map(np.array([0.4]), depth, tf, 0.5, 5.0, FOV)
    seen = v._map > 0
    assert seen.sum() > 5000
    # first observation: conf' = new, value' = values
    assert np.allclose(v._value_map[seen, 0], 0.4, atol=1e-7) and v._value_map.dtype == np.float64
    conf1 = v._map.copy()
    # same observation twice: unchanged
    v.update_map(np.array([0.4]), depth, tf, 0.5, 5.0, FOV)
    assert np.allclose(v._map, conf1, atol=1e-6) and np.allclose(v._value_map[seen, 0], 0.4, atol=1e-7)
    # different value at equal confidence: arithmetic mean
    v.update_map(np.array([0.2]), depth, tf, 0.5, 5.0, FOV)
    assert np.allclose(v._value_map[seen, 0], 0.3, atol=1e-6)
    assert (v._value_map[~seen] == 0).all() and (v._map[~seen] == 0).all()   # 0/0 -> 0


def test_fusion_decision_threshold_and_max_confidence():
    v = RefValueMap(1, use_max_confidence=True)
    depth = np.ones((480, 640), np.float32)
    v.update_map(np.array([0.4]), depth, pose_to_tf((0, 0, 0.88), 0.0), 0.5, 5.0, FOV)
    conf1, val1 = v._map.copy(), v._value_map.copy()
    # look at the same area from a rotated pose: cells where new <= old keep their value; new > old are replaced
    v.update_map(np.array([0.1]), depth, pose_to_tf((0, 0, 0.88), 0.5), 0.5, 5.0, FOV)
    both = (conf1 > 0)
    replaced = v._value_map[..., 0] == np.float32(0.1)
    assert (v._map[both] >= conf1[both]).all()
    assert (v._map[replaced & both] > conf1[replaced & both]).all()
    kept = both & ~replaced
    assert np.array_equal(v._value_map[kept], val1[kept]) and v._value_map.dtype == np.float32
    # new < 0.35 and new < old is ignored outright in the weighted mode too
    w = RefValueMap(1, use_max_confidence=False)
    w.update_map(np.array([0.4]), depth, pose_to_tf((0, 0, 0.88), 0.0), 0.5, 5.0, FOV)
    c1, v1 = w._map.copy(), w._value_map.copy()
    w.update_map(np.array([0.1]), depth, pose_to_tf((0, 0, 0.88), 0.62), 0.5, 5.0, FOV)
    new_only = RefValueMap(1, use_max_confidence=False)
    new_only.update_map(np.array([0.1]), depth, pose_to_tf((0, 0, 0.88), 0.62), 0.5, 5.0, FOV)
    n = new_only._map
    ignored = (n > 0) & (n < 0.35) & (n < c1)
    assert ignored.sum() > 100
    assert np.array_equal(w._map[ignored], c1[ignored]) and np.allclose(w._value_map[ignored], v1[ignored])


def test_index_conventions():
    b = RefBaseMap()
    # rint (half-even) for the obstacle map: x=0.025 m -> 0.5 cells -> 0 ; x=0.075 -> 1.5 -> 2
    px = b._xy_to_px(np.array([[0.025, 0.0], [0.075, 0.0], [1.0, -2.0]]))
    assert px.tolist() == [[500, 500], [500, 502], [540, 520]]
    assert np.allclose(b._px_to_xy(px.astype(float))[2], [1.0, -2.0])
    # truncation for the value map: camera at x = 0.049 m and -0.049 m both land on the origin row
    v = RefValueMap(1, use_max_confidence=False)
    depth = np.ones((480, 640), np.float32)
    v.update_map(np.array([0.3]), depth, pose_to_tf((0.049, -0.049, 0.88), 0.0), 0.5, 5.0, FOV)
    a = v._map.copy()
    v2 = RefValueMap(1, use_max_confidence=False)
    v2.update_map(np.array([0.3]), depth, pose_to_tf((-0.049, 0.049, 0.88), 0.0), 0.5, 5.0, FOV)
    assert np.array_equal(a, v2._map) and a[500, 500] > 0
    # map row = S//2 + x*ppm, col = S//2 - y*ppm
    v3 = RefValueMap(1, use_max_confidence=False)
    v3.update_map(np.array([0.3]), depth, pose_to_tf((2.0, 1.0, 0.88), 0.0), 0.5, 5.0, FOV)
    assert v3._map[540, 480] > 0 and v3._map[539, 480] == 0   # apex at the camera cell, cone towards +row
    # sort_waypoints: -1 for never-seen, descending order
    pts, vals = v3.sort_waypoints(np.array([[10.0, 10.0], [4.0, 1.0], [3.0, 1.0]]), 0.5)
    assert vals[-1] == -1 and pts[-1].tolist() == [10.0, 10.0] and vals[0] >= vals[1] > 0


def test_error_conventions():
    v = RefValueMap(2, use_max_confidence=False)
    depth = np.ones((480, 640), np.float32)
    with pytest.raises(AssertionError, match="Incorrect number of values"):
        v.update_map(np.array([0.3]), depth, pose_to_tf((0, 0, 0.88), 0.0), 0.5, 5.0, FOV)
    with pytest.raises(AssertionError, match="outside the image"):
        v.update_map(np.array([0.3, 0.2]), depth, pose_to_tf((25.1, 0, 0.88), 0.0), 0.5, 5.0, FOV)
    v.update_map(np.array([0.3, 0.2]), depth, pose_to_tf((0, 0, 0.88), 0.0), 0.5, 5.0, FOV)
    with pytest.raises(AssertionError, match="reduction function"):
        v.sort_waypoints(np.array([[1.0, 0.0]]), 0.5)

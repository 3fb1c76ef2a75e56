"""Known-answer pins for oracle/cvport.c, derived by hand from the published OpenCV 4.5.x rasterisation rules
(the reference has no golden vectors for this path -- SURVEY.md 8c; these are the substitute pins)."""
import numpy as np

from oracle import cv


def test_line8_bresenham_known_pixels():
    img = np.zeros((8, 8), np.uint8)
    cv.lib().cvp_line8(img, 8, 8, 0, 0, 7, 3, 1)
    ys = [int(np.where(img[:, x])[0][0]) for x in range(8)]
    # err = dx - 2dy = 1; minor step when err < 0
    assert ys == [0, 0, 1, 1, 2, 2, 3, 3]
    assert img.sum() == 8
    # right-to-left input is walked left-to-right: same pixels
    img2 = np.zeros((8, 8), np.uint8)
    cv.lib().cvp_line8(img2, 8, 8, 7, 3, 0, 0, 1)
    assert np.array_equal(img, img2)


def test_fill_poly_square_and_triangle():
    img = np.zeros((10, 10), np.uint8)
    cv.drawContours(img, [np.array([[2, 2], [6, 2], [6, 5], [2, 5]])], -1, 1, -1)
    want = np.zeros((10, 10), np.uint8)
    want[2:6, 2:7] = 1
    assert np.array_equal(img, want)
    tri = np.zeros((10, 10), np.uint8)
    cv.drawContours(tri, [np.array([[0, 0], [8, 0], [0, 8]])], -1, 1, -1)
    # right-isoceles triangle incl. its Bresenham hypotenuse: row y covers x = 0..8-y
    for y in range(9):
        assert tri[y].sum() == 9 - y and tri[y, : 9 - y].all()


def test_fill_poly_even_odd_two_contours():
    img = np.zeros((12, 12), np.uint8)
    outer = np.array([[1, 1], [10, 1], [10, 10], [1, 10]])
    inner = np.array([[4, 4], [7, 4], [7, 7], [4, 7]])
    cv.drawContours(img, [outer, inner], -1, 1, -1)
    # even-odd: the strict interior of the inner square is a hole, its boundary is drawn
    assert img[5, 5] == 0 and img[6, 6] == 0 and img[4, 4] == 1 and img[2, 2] == 1


def test_fill_poly_clips_outside_points():
    img = np.zeros((6, 6), np.uint8)
    cv.drawContours(img, [np.array([[-5, -5], [20, -5], [20, 20], [-5, 20]])], -1, 1, -1)
    assert img.all()


def test_circle_radius_1_2_10():
    img = np.zeros((5, 5), np.uint8)
    cv.circle(img, (2, 2), 1, 255, -1)
    # midpoint iteration 1: rows +-0 get half-width 1, rows +-1 get half-width 0; then dx drops below dy -> a plus
    assert np.array_equal(img > 0, np.array([[0, 0, 0, 0, 0], [0, 0, 1, 0, 0], [0, 1, 1, 1, 0], [0, 0, 1, 0, 0],
                                             [0, 0, 0, 0, 0]], bool))
    d = np.zeros((21, 21), np.uint8)
    cv.circle(d, (10, 10), 10, 255, -1)
    m = d > 0
    assert m[10].all() and m[:, 10].all() and np.array_equal(m, m.T) and np.array_equal(m, m[::-1]) and not m[0, 0]
    yy, xx = np.mgrid[-10:11, -10:11]
    r2 = xx * xx + yy * yy
    assert m[r2 <= 90].all() and not m[r2 >= 121].any()
    # disc drawn off-centre in a clipped crop stays clipped to the crop (img_utils.py:246-253)
    c = np.zeros((15, 21), np.uint8)
    cv.circle(c, (10, 10), 10, 255, -1)
    assert np.array_equal(c > 0, m[:15])


def test_dilate_rect_border_ignored():
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 1
    img[0, 0] = 1
    out = cv.dilate(img, np.ones((7, 7), np.uint8))
    want = np.zeros((9, 9), np.uint8)
    want[1:8, 1:8] = 1
    want[0:4, 0:4] = 1
    assert np.array_equal(out, want)
    assert np.array_equal(cv.dilate(img, np.ones((3, 3), np.uint8))[3:6, 3:6], np.ones((3, 3), np.uint8))


def test_find_contours_simple_and_none():
    img = np.zeros((8, 8), np.uint8)
    img[2:5, 3:7] = 1
    cs, _ = cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    assert len(cs) == 1
    # outer border starts at the top-left pixel and runs counter-clockwise (y down): TL, BL, BR, TR
    assert cs[0].reshape(-1, 2).tolist() == [[3, 2], [3, 4], [6, 4], [6, 2]]
    cn, _ = cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_NONE)
    assert len(cn[0]) == 10 and cn[0].reshape(-1, 2)[0].tolist() == [3, 2]
    assert cv.contourArea(cs[0]) == 6.0  # polygon through pixel centres: 3 x 2


def test_find_contours_order_holes_and_single_pixels():
    img = np.zeros((12, 12), np.uint8)
    img[1, 1] = 1                       # single pixel, found first
    img[3:9, 3:9] = 1
    img[5:7, 5:7] = 0                   # hole
    img[10, 8] = 1                      # found last
    ext, _ = cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    assert len(ext) == 3
    assert ext[0].reshape(-1, 2).tolist() == [[8, 10]]   # last found is returned first
    assert ext[2].reshape(-1, 2).tolist() == [[1, 1]]
    lst, _ = cv.findContours(img, cv.RETR_LIST, cv.CHAIN_APPROX_SIMPLE)
    assert len(lst) == 4                                  # + the hole border
    tree, _ = cv.findContours(img, cv.RETR_TREE, cv.CHAIN_APPROX_SIMPLE)
    assert len(tree) == 4
    # a component nested in the hole is external-invisible
    img[5, 5] = 1
    ext2, _ = cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    assert len(ext2) == 3
    # image untouched, input with values > 1 treated as nonzero
    img2 = img * 200
    assert len(cv.findContours(img2, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)[0]) == 3


def test_find_contours_touching_image_border():
    img = np.ones((5, 7), np.uint8)
    cs, _ = cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    assert len(cs) == 1 and cs[0].reshape(-1, 2).tolist() == [[0, 0], [0, 4], [6, 4], [6, 0]]


def test_fill_of_external_contour_fills_holes():
    rng = np.random.default_rng(0)
    img = (rng.uniform(size=(40, 40)) < 0.6).astype(np.uint8)
    cs, _ = cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    filled = np.zeros_like(img)
    cv.drawContours(filled, cs, -1, 1, -1)
    # every foreground pixel is inside the fill of some external contour
    assert (filled[img > 0] == 1).all()
    # background reachable from the image border by 4-moves is never filled
    from scipy import ndimage
    lab, _ = ndimage.label(np.pad(img == 0, 1, constant_values=True))
    outside = (lab == lab[0, 0])[1:-1, 1:-1]
    assert (filled[outside] == 0).all()
    assert (filled[(img == 0) & ~outside] == 1).all()


def test_point_polygon_test_and_convexity():
    sq = np.array([[2, 2], [2, 8], [8, 8], [8, 2]]).reshape(-1, 1, 2)
    assert cv.pointPolygonTest(sq, (5, 5), True) == 3.0
    assert cv.pointPolygonTest(sq, (2, 5), True) == 0.0
    assert cv.pointPolygonTest(sq, (0, 5), True) == -2.0
    assert abs(cv.pointPolygonTest(sq, (11, 12), True) + 5.0) < 1e-12
    assert cv.isContourConvex(sq)
    assert not cv.isContourConvex(np.array([[0, 0], [0, 4], [2, 1], [4, 4], [4, 0]]))
    assert not cv.isContourConvex(np.array([[0, 0], [0, 2], [0, 4], [4, 4], [4, 0]]))  # collinear triple


def test_warp_affine_identity_shift_and_quarter_turn():
    rng = np.random.default_rng(1)
    src = rng.uniform(size=(11, 11))
    ident = np.array([[1.0, 0, 0], [0, 1, 0]])
    assert np.array_equal(cv.warpAffine(src, ident, (11, 11)), src)
    shift = np.array([[1.0, 0, 2], [0, 1, 1]])
    out = cv.warpAffine(src, shift, (11, 11))
    assert np.array_equal(out[1:, 2:], src[:-1, :-2]) and (out[0] == 0).all() and (out[:, :2] == 0).all()
    # half-pixel shift: dst(x) = 0.5*src(x-1) + 0.5*src(x) exactly (weights are multiples of 1/32)
    half = cv.warpAffine(src, np.array([[1.0, 0, 0.5], [0, 1, 0]]), (11, 11))
    assert np.allclose(half[:, 1:], 0.5 * src[:, :-1] + 0.5 * src[:, 1:], rtol=0, atol=1e-15)
    # 90 degree rotation about the centre is a pure index permutation
    M = cv.getRotationMatrix2D((5, 5), 90.0, 1.0)
    rot = cv.warpAffine(src, M, (11, 11))
    assert np.allclose(rot, np.rot90(src, 1), rtol=0, atol=1e-15)


def test_ellipse_sector_polygon_fov79():
    fov = np.deg2rad(79)
    a0, a1 = -np.rad2deg(fov) / 2 + 90, np.rad2deg(fov) / 2 + 90
    assert abs(a0 - 50.5) < 1e-9 and abs(a1 - 129.5) < 1e-9      # cvRound half-even -> 50 and 130
    poly = cv.ellipse_polygon((100, 100), (100, 100), 0, a0, a1)
    assert len(poly) == 18                                         # 17 arc vertices (5 deg steps) + the centre
    assert poly[-1].tolist() == [100 << 16, 100 << 16]
    assert poly[8].tolist() == [100 << 16, 200 << 16]              # 90 deg vertex straight ahead
    assert abs(poly[0, 0] / 65536 - (100 + 100 * np.cos(np.deg2rad(50)))) < 1e-4
    img = np.zeros((201, 201))
    cv.ellipse(img, (100, 100), (100, 100), 0, a0, a1, 1, -1)
    assert img[100, 100] == 1 and img[200, 100] == 1 and img[99].sum() == 0 and img[150, 100] == 1
    # near-symmetric about the optical axis; the left-to-right line walk and truncating 16.16 slopes make the raster
    # differ from its mirror image only in isolated boundary pixels
    assert (img != img[:, ::-1]).sum() <= 2 * 201
    assert img[150, 100 + 60] == 0 and img[150, 100 + 41] == 1     # tan(40 deg)*50 = 41.95


def test_blur3x3_is_a_3x3_any_for_binary_255():
    rng = np.random.default_rng(2)
    m = (rng.uniform(size=(20, 20)) < 0.1).astype(np.uint8) * 255
    out = cv.blur(m, (3, 3))
    from scipy import ndimage
    assert np.array_equal(out > 0, ndimage.maximum_filter(m, size=3, mode="mirror") > 0)


def test_thick_polyline_covers_thin_line_and_is_wider():
    img = np.ones((40, 40), np.uint8)
    cv.polylines(img, np.array([[[5, 5], [30, 20]]], np.int32), False, 0, 2)
    thin = np.zeros((40, 40), np.uint8)
    cv.lib().cvp_line8(thin, 40, 40, 5, 5, 30, 20, 1)
    assert (img[thin > 0] == 0).all()
    assert (img == 0).sum() > thin.sum()
    # the cut separates the two sides: top-right and bottom-left corners are in different 4-connected components
    from scipy import ndimage
    lab, n = ndimage.label(img[8:18, 11:25])  # a window the band crosses completely
    assert n >= 2


# ------------------------------------------------------------------------------------------ cvport.c beside the REAL cv2
def _real_cv2_or_skip():
    """The real wheel, if this machine has one (tools/verify_with_real_vlfm.md); with VLFM_REAL_CV2=1 its absence already
    failed at ``from oracle import cv``."""
    import pytest

    try:
        return cv.import_real_cv2()
    except ImportError:
        pytest.skip("no real cv2 on this machine (opencv-python==4.5.5.64 is the reference's pin)")


def _blobs(rng, shape=(160, 200)):
    img = np.zeros(shape, np.uint8)
    for _ in range(int(rng.integers(2, 7))):
        cy, cx = rng.integers(10, shape[0] - 10), rng.integers(10, shape[1] - 10)
        ry, rx = rng.integers(2, 25), rng.integers(2, 25)
        img[max(cy - ry, 0):cy + ry, max(cx - rx, 0):cx + rx] = 1
    holes = rng.integers(0, 2, shape).astype(np.uint8)
    return img & (holes | (rng.random(shape) < 0.97).astype(np.uint8))


def test_cvport_equals_real_cv2_on_the_entry_points_of_the_path():
    """Side by side on seeded random inputs, bit for bit: the sector mask and rotation of value_map.py:260,325-334, the filled
    polygons of img_utils.py / obstacle_map.py:164, the dilation of obstacle_map.py:117, the contour calls of :128-169."""
    real = _real_cv2_or_skip()
    S = cv.STANDIN
    rng = np.random.default_rng(20240926)
    for it in range(40):
        size = int(rng.choice([101, 201, 401]))
        c = size // 2
        fov = float(rng.uniform(40, 110))
        for dtype in (np.uint8, np.float64):
            a, b = np.zeros((size, size), dtype), np.zeros((size, size), dtype)
            args = ((c, c), (c, c), 0, -fov / 2 + 90, fov / 2 + 90, 1, -1)
            S["ellipse"](a, *args); real.ellipse(b, *args)
            assert np.array_equal(a, b), ("ellipse", size, fov, dtype)
        angle = float(rng.uniform(-360, 360))
        M0, M1 = S["getRotationMatrix2D"]((c, c), angle, 1.0), real.getRotationMatrix2D((c, c), angle, 1.0)
        assert np.array_equal(M0, M1), ("getRotationMatrix2D", angle)
        src = rng.random((size, size)) * (rng.random((size, size)) < 0.5)
        assert np.array_equal(S["warpAffine"](src, M0, (size, size)), real.warpAffine(src, M1, (size, size))), ("warpAffine", angle)
        poly = rng.integers(-20, size + 20, (int(rng.integers(3, 40)), 2)).astype(np.int32)
        a, b = np.zeros((size, size), np.uint8), np.zeros((size, size), np.uint8)
        S["drawContours"](a, [poly], -1, 1, -1); real.drawContours(b, [poly], -1, 1, -1)
        assert np.array_equal(a, b), ("drawContours", it)
        img = _blobs(rng)
        k = int(rng.choice([3, 7, 9]))
        kern = np.ones((k, k), np.uint8)
        assert np.array_equal(S["dilate"](img, kern, iterations=1), real.dilate(img, kern, iterations=1)), ("dilate", k)
        for mode, method in ((0, 1), (0, 2), (1, 1), (1, 2)):   # RETR_EXTERNAL / RETR_LIST x CHAIN_APPROX_NONE / SIMPLE
            c0, _ = S["findContours"](img.copy(), mode, method)
            c1 = real.findContours(img.copy(), mode, method)[-2]
            assert len(c0) == len(c1), ("findContours count", mode, method)
            for p, q in zip(c0, c1):
                assert np.array_equal(np.asarray(p), np.asarray(q)), ("findContours", mode, method)
                assert S["contourArea"](p) == real.contourArea(q)
                pt = (float(rng.integers(0, img.shape[1])), float(rng.integers(0, img.shape[0])))
                assert S["pointPolygonTest"](p, pt, False) == real.pointPolygonTest(q, pt, False)
                assert S["pointPolygonTest"](p, pt, True) == real.pointPolygonTest(q, pt, True)
        cen = (int(rng.integers(0, size)), int(rng.integers(0, size)))
        a, b = np.zeros((size, size), np.uint8), np.zeros((size, size), np.uint8)
        r = int(rng.integers(1, 30))
        S["circle"](a, cen, r, 1, -1); real.circle(b, cen, r, 1, -1)
        assert np.array_equal(a, b), ("circle", cen, r)

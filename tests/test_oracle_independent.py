"""Second opinions on oracle/cvport.c from code that shares nothing with it: the stand-ins for OpenCV are
restatements by one author, and the device code (raster.h, border_parallel.h) is compared with THEM -- an error shared by both is
invisible to the parity tests.  This file checks the stand-ins against

  * scipy.ndimage (binary_dilation / binary_erosion / label / binary_fill_holes): morphology, component counts, hole filling;
  * a border follower written here from the PAPER (Suzuki & Abe 1985, Algorithm 1: raster scan, NBD labels, clockwise first search,
    counter-clockwise tracing, the "(i3, j3 + 1) was examined" rule for -NBD) in plain Python: point sequences and contour order on
    ALL 65 536 binary 4 x 4 images and on random 7 x 9 images, CHAIN_APPROX_NONE and CHAIN_APPROX_SIMPLE;
  * PIL.ImageDraw (polygon, pieslice) as a LOOSER opinion on fill membership.  Known, enumerated rule differences: OpenCV draws the
    8-connected Bresenham outline of every edge into the fill (PIL's scanline rule decides boundary pixels by its own centre
    sampling), so agreement is demanded only for pixels farther than one pixel from every edge; what the stand-in fills must
    contain everything PIL fills strictly inside and nothing strictly outside.

Nothing here can prove that cv2 == cvport (cv2 is not installable in this image; tools/verify_with_real_vlfm.md is the recipe for
that); it removes the "both sides were written from the same recollection" failure mode for the rules these libraries share."""
import itertools

import numpy as np
import pytest
from scipy import ndimage

from oracle import cv


# ----------------------------------------------------------------------------------------------------------- morphology
@pytest.mark.parametrize("k", [3, 5, 7, 9])
def test_dilate_equals_scipy_binary_dilation(k):
    rng = np.random.default_rng(100 + k)
    st = np.ones((k, k), bool)
    for density, shape in itertools.product((0.002, 0.05, 0.4), ((37, 53), (64, 64), (5, 4), (1, 30))):
        img = (rng.uniform(size=shape) < density).astype(np.uint8)
        img[0, 0] = 1                                     # the border rule: taps outside the image are ignored
        got = cv.dilate(img, np.ones((k, k), np.uint8))
        want = ndimage.binary_dilation(img.astype(bool), structure=st, border_value=0)
        assert np.array_equal(got.astype(bool), want), (k, density, shape)


def test_erode_equals_scipy_binary_erosion_with_ignored_border():
    rng = np.random.default_rng(5)
    for it in (1, 2, 5):
        img = (rng.uniform(size=(60, 70)) < 0.93).astype(np.uint8)
        got = cv.erode(img, None, iterations=it)          # 3 x 3, taps outside the image do not constrain (OpenCV's +inf border)
        want = ndimage.binary_erosion(img.astype(bool), structure=np.ones((3, 3), bool), iterations=it, border_value=1)
        assert np.array_equal(got.astype(bool), want), it


# ------------------------------------------------------------------------------------------- contours vs scipy's components
def _top_level_components(img):
    """8-connected foreground components that touch the OUTER background (the 4-connected background region that contains the frame)."""
    fg = np.pad(img.astype(bool), 1)
    lab_f, nf = ndimage.label(fg, structure=np.ones((3, 3), int))
    lab_b, _ = ndimage.label(~fg)                         # 4-connected background
    outer = lab_b == lab_b[0, 0]
    near_outer = ndimage.binary_dilation(outer, structure=ndimage.generate_binary_structure(2, 1))   # 4-neighbours of the outer region
    ids = np.unique(lab_f[near_outer & fg])
    return lab_f[1:-1, 1:-1], [int(i) for i in ids if i > 0]


@pytest.mark.parametrize("seed", range(6))
def test_external_contours_match_scipy_components_and_hole_filling(seed):
    rng = np.random.default_rng(seed)
    if seed < 3:                                          # salt-and-pepper: every pathological touching case
        img = (rng.uniform(size=(48, 61)) < (0.35, 0.5, 0.62)[seed]).astype(np.uint8)
    else:                                                 # blobs with holes and nested islands
        img = (ndimage.gaussian_filter(rng.normal(size=(90, 120)), 2.0 + seed) > 0.02).astype(np.uint8)
        img[40:50, 50:70] = 1
        img[43:47, 55:65] = 0
        img[45, 60] = 1                                   # an island inside a hole: invisible to RETR_EXTERNAL
    lab, top = _top_level_components(img)
    cs, _ = cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_NONE)
    assert len(cs) == len(top)
    # one contour per top-level component, made of that component's pixels that have a 4-neighbour in the outer background
    fgp = np.pad(img.astype(bool), 1)
    lab_b, _ = ndimage.label(~fgp)
    outer = lab_b == lab_b[0, 0]
    near_outer = ndimage.binary_dilation(outer, structure=ndimage.generate_binary_structure(2, 1))[1:-1, 1:-1]
    seen = set()
    for c in cs:
        pts = c.reshape(-1, 2)
        ids = {int(lab[y, x]) for x, y in pts}
        assert len(ids) == 1 and ids <= set(top)
        seen |= ids
        want = {(int(x), int(y)) for y, x in zip(*np.nonzero((lab == next(iter(ids))) & near_outer))}
        assert {(int(x), int(y)) for x, y in pts} == want
    assert seen == set(top)
    # filling the external contours = the top-level components with every hole (and what sits in it) filled
    filled = np.zeros_like(img)
    cv.drawContours(filled, cs, -1, 1, -1)
    want = ndimage.binary_fill_holes(np.isin(lab, top))   # default structure: 4-connected background, as the border follower sees it
    assert np.array_equal(filled.astype(bool), want)
    # CHAIN_APPROX_SIMPLE describes the same polygons: same fill
    cs2, _ = cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    filled2 = np.zeros_like(img)
    cv.drawContours(filled2, cs2, -1, 1, -1)
    assert np.array_equal(filled2, filled)


# -------------------------------------------------------------------------------- Suzuki & Abe 1985, Algorithm 1, from the paper
# neighbour k of a pixel, COUNTER-clockwise in image coordinates with the row index growing downwards, starting east:
# E, NE, N, NW, W, SW, S, SE.  "Clockwise" is the reverse walk.
_NB = [(0, 1), (-1, 1), (-1, 0), (-1, -1), (0, -1), (1, -1), (1, 0), (1, 1)]


def _suzuki_abe(img):
    """All borders of a binary image: list of (is_hole, parent_nbd, nbd, [(row, col), ...]) in the order the raster scan finds them.
    Row / column indices refer to the unpadded image."""
    f = np.pad((np.asarray(img) != 0).astype(np.int64), 1)
    rows, cols = f.shape
    nbd = 1
    kind = {1: True}          # border number -> is_hole; the frame counts as a hole border
    parent = {1: 0}
    out = []
    for i in range(1, rows - 1):
        lnbd = 1
        for j in range(1, cols - 1):
            v = f[i, j]
            if v == 0:
                continue
            start = None
            if v == 1 and f[i, j - 1] == 0:               # (1a) outer border
                nbd += 1
                start, is_hole = (i, j - 1), False
            elif v >= 1 and f[i, j + 1] == 0:             # (1b) hole border
                nbd += 1
                start, is_hole = (i, j + 1), True
                if v > 1:
                    lnbd = v
            if start is not None:
                # (2) parent from Table 1 of the paper
                if is_hole == kind[lnbd]:
                    parent[nbd] = parent[lnbd]
                else:
                    parent[nbd] = lnbd
                kind[nbd] = is_hole
                pts = []
                k0 = _NB.index((start[0] - i, start[1] - j))
                # (3.1) clockwise from (i2, j2) around (i, j): first nonzero pixel
                first = None
                for s in range(8):
                    k = (k0 - s) % 8
                    if f[i + _NB[k][0], j + _NB[k][1]] != 0:
                        first = (i + _NB[k][0], j + _NB[k][1])
                        break
                if first is None:
                    f[i, j] = -nbd
                    pts.append((i - 1, j - 1))
                else:
                    i2, j2 = first                         # (3.2)
                    i3, j3 = i, j
                    while True:
                        # (3.3) counter-clockwise around (i3, j3), starting from the element after (i2, j2)
                        kk = _NB.index((i2 - i3, j2 - j3))
                        east_was_zero = False
                        nxt = None
                        for s in range(1, 9):
                            k = (kk + s) % 8
                            y, x = i3 + _NB[k][0], j3 + _NB[k][1]
                            if f[y, x] != 0:
                                nxt = (y, x)
                                break
                            if k == 0:
                                east_was_zero = True       # (i3, j3 + 1) is a 0-pixel examined in this step
                        # (3.4)
                        if east_was_zero:
                            f[i3, j3] = -nbd
                        elif f[i3, j3] == 1:
                            f[i3, j3] = nbd
                        pts.append((i3 - 1, j3 - 1))
                        # (3.5)
                        if nxt == (i, j) and (i3, j3) == first:
                            break
                        i2, j2 = i3, j3
                        i3, j3 = nxt
                out.append((is_hole, parent[nbd], nbd, pts))
            # (4)
            if f[i, j] != 1:
                lnbd = abs(int(f[i, j]))
    return out


def _external_from_paper(img):
    """RETR_EXTERNAL in OpenCV's list order (the border found LAST comes first), points as (x, y)."""
    ext = [pts for is_hole, par, _, pts in _suzuki_abe(img) if not is_hole and par == 1]
    return [[(c, r) for r, c in pts] for pts in reversed(ext)]


def _simple(points):
    """CHAIN_APPROX_SIMPLE of a closed 8-connected point sequence, written from its definition: a point is kept when the step that
    arrives at it and the step that leaves it differ (run end points of horizontal, vertical and diagonal runs); a contour of one
    point is that point; when every step is the same (impossible for a closed curve of > 1 points) nothing would be dropped."""
    n = len(points)
    if n == 1:
        return list(points)
    keep = []
    for t in range(n):
        px, py = points[t - 1]
        cx, cy = points[t]
        nx, ny = points[(t + 1) % n]
        if (cx - px, cy - py) != (nx - cx, ny - cy):
            keep.append((cx, cy))
    return keep


def _as_lists(contours):
    return [[(int(x), int(y)) for x, y in c.reshape(-1, 2)] for c in contours]


def test_find_contours_equals_paper_algorithm_on_every_4x4_image():
    """65 536 images: every configuration of touching, nesting and single pixels that fits in 4 x 4."""
    bad = 0
    for code in range(1 << 16):
        img = ((code >> np.arange(16)) & 1).astype(np.uint8).reshape(4, 4)
        want = _external_from_paper(img)
        got = _as_lists(cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_NONE)[0])
        if got != want:
            bad += 1
            assert bad < 1, (code, img.tolist(), got, want)
        if code % 7 == 0:                                 # (the SIMPLE form on a seventh of them: it is a pure function of NONE)
            got_s = _as_lists(cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)[0])
            assert got_s == [_simple(p) for p in want], (code, img.tolist())


@pytest.mark.parametrize("density", [0.3, 0.5, 0.7])
def test_find_contours_equals_paper_algorithm_on_random_images(density):
    rng = np.random.default_rng(int(density * 100))
    for _ in range(400):
        img = (rng.uniform(size=(7, 9)) < density).astype(np.uint8)
        want = _external_from_paper(img)
        assert _as_lists(cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_NONE)[0]) == want, img.tolist()
        assert _as_lists(cv.findContours(img, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)[0]) == [_simple(p) for p in want], img.tolist()
    # RETR_LIST: every border of the paper's algorithm, holes included, same reversed order
    for _ in range(100):
        img = (ndimage.gaussian_filter(rng.normal(size=(24, 30)), 1.5) > 0).astype(np.uint8)
        allb = [[(c, r) for r, c in pts] for _, _, _, pts in reversed(_suzuki_abe(img))]
        assert _as_lists(cv.findContours(img, cv.RETR_LIST, cv.CHAIN_APPROX_NONE)[0]) == allb


# ------------------------------------------------------------------------------------------------------ fills vs PIL (looser)
def _edge_distance(shape, poly):
    """Distance of every pixel centre to the nearest polygon edge."""
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]].astype(np.float64)
    d = np.full(shape, np.inf)
    p = np.asarray(poly, np.float64)
    for a, b in zip(p, np.roll(p, -1, axis=0)):
        ab = b - a
        den = float(ab @ ab)
        t = np.zeros(shape) if den == 0 else np.clip(((xx - a[0]) * ab[0] + (yy - a[1]) * ab[1]) / den, 0, 1)
        d = np.minimum(d, np.hypot(xx - (a[0] + t * ab[0]), yy - (a[1] + t * ab[1])))
    return d


def test_fill_poly_agrees_with_pil_away_from_the_edges():
    from PIL import Image, ImageDraw

    rng = np.random.default_rng(11)
    for trial in range(200):
        n = int(rng.integers(3, 9))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = rng.uniform(6, 28, n)
        poly = np.stack([32 + rad * np.cos(ang), 32 + rad * np.sin(ang)], 1).round().astype(np.int32)     # star-shaped: simple polygon
        ours = np.zeros((64, 64), np.uint8)
        cv.drawContours(ours, [poly], -1, 1, -1)
        im = Image.new("L", (64, 64), 0)
        ImageDraw.Draw(im).polygon([tuple(map(int, p)) for p in poly], fill=1, outline=1)
        pil = np.asarray(im)
        far = _edge_distance((64, 64), poly) > 1.0
        assert np.array_equal(ours[far], pil[far]), (trial, poly.tolist())
        # the stand-in's fill contains the 8-connected outline of every edge (OpenCV draws it), PIL's outline is a subset check
        edge = np.zeros((64, 64), np.uint8)
        for (x0, y0), (x1, y1) in zip(poly, np.roll(poly, -1, axis=0)):
            cv.lib().cvp_line8(edge, 64, 64, int(x0), int(y0), int(x1), int(y1), 1)
        assert (ours[edge > 0] == 1).all()


def test_ellipse_sector_agrees_with_pil_pieslice_away_from_the_boundary():
    """The fog-of-war / blank-cone sector (value_map.py:321-335): cv2.ellipse with a 5-degree polygon vs PIL's analytic pie slice.
    The polygon lies INSIDE the true circle by up to r (1 - cos 2.5 deg) = 0.1 px at r = 100, so the two may differ only within
    ~1.5 px of the arc and of the two radii."""
    from PIL import Image, ImageDraw

    for fov_deg, r in ((79.0, 100), (60.0, 57), (110.0, 100)):
        T = 2 * r + 1
        ours = np.zeros((T, T), np.uint8)
        a0, a1 = 90 - fov_deg / 2, 90 + fov_deg / 2
        cv.ellipse(ours, (r, r), (r, r), 0, a0, a1, 1, -1)
        im = Image.new("L", (T, T), 0)
        ImageDraw.Draw(im).pieslice([0, 0, 2 * r, 2 * r], round(a0), round(a1), fill=1)      # cv2.ellipse rounds its angles too
        pil = np.asarray(im)
        yy, xx = np.mgrid[0:T, 0:T]
        rho = np.hypot(xx - r, yy - r)
        th = np.degrees(np.arctan2(yy - r, xx - r))
        margin_ang = np.degrees(2.0 / np.maximum(rho, 1.0))
        safe = (rho < r - 2.0) & (rho > 3.0) & (np.abs(th - round(a0)) > margin_ang) & (np.abs(th - round(a1)) > margin_ang)
        safe |= rho > r + 2.0
        assert np.array_equal(ours[safe], pil[safe]), (fov_deg, r)
        assert abs(int(ours.sum()) - int(pil.sum())) <= 0.03 * pil.sum()


# ----------------------------------------------------------------------------------------------------------- rotation, disc, blur, contour scalars
@pytest.mark.parametrize("deg", [0.0, 17.3, 90.0, 133.0, -61.5, 180.0])
def test_warp_affine_agrees_with_exact_bilinear_resampling_within_its_fixed_point_grid(deg):
    """rotate_image (img_utils.py:9-28) = getRotationMatrix2D + warpAffine(INTER_LINEAR).  OpenCV evaluates the inverse map in fixed
    point (coordinates on a 1/32-pixel grid, bilinear weights in 1/32768), so it cannot be bit-equal to an exact resampler -- but it
    must be the SAME MAP: same centre, same sense of rotation (positive = counter-clockwise on screen), zero outside.  Compared with
    scipy.ndimage.map_coordinates(order=1) fed with the analytic inverse rotation, on a smooth image: the difference is bounded by
    the coordinate grid (1/32 px) times the image's largest gradient -- a wrong centre or sign would be off by whole pixels."""
    h = w = 101
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    src = 0.5 + 0.25 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 0.002 * xx               # smooth, asymmetric
    c = (w // 2, h // 2)
    got = cv.warpAffine(src, cv.getRotationMatrix2D(c, deg, 1.0), (w, h))
    a = np.deg2rad(deg)
    # forward map of cv2.getRotationMatrix2D: x' = cos a (x - cx) + sin a (y - cy) + cx, y' = -sin a (x - cx) + cos a (y - cy) + cy;
    # the sampled source point of destination (x', y') is its inverse
    dx, dy = xx - c[0], yy - c[1]
    sx = np.cos(a) * dx - np.sin(a) * dy + c[0]
    sy = np.sin(a) * dx + np.cos(a) * dy + c[1]
    want = ndimage.map_coordinates(src, [sy, sx], order=1, mode="constant", cval=0.0)
    inside = (sx > 1) & (sx < w - 2) & (sy > 1) & (sy < h - 2)
    grad = max(np.abs(np.diff(src, axis=0)).max(), np.abs(np.diff(src, axis=1)).max())
    assert inside.sum() > 5000
    assert np.abs(got - want)[inside].max() <= 2.0 * grad / 32 + 1e-4, deg
    far = (sx < -1) | (sx > w) | (sy < -1) | (sy > h)
    assert far.sum() == 0 or np.all(got[far] == 0.0)


@pytest.mark.parametrize("r", [1, 2, 5, 10, 17])
def test_filled_circle_is_the_analytic_disc_up_to_its_boundary_ring(r):
    """cv2.circle(..., thickness=-1) behind pixel_value_within_radius (img_utils.py:247-253): every pixel strictly inside the
    radius is set, none farther than radius + 1, and the shape has the disc's eightfold symmetry."""
    n = 2 * r + 9
    img = cv.circle(np.zeros((n, n), np.uint8), (n // 2, n // 2), r, 1, -1) > 0
    yy, xx = np.mgrid[0:n, 0:n]
    d = np.hypot(xx - n // 2, yy - n // 2)
    assert img[d <= r - 0.75].all() and not img[d >= r + 1].any()
    assert img[n // 2, n // 2 + r] and img[n // 2 + r, n // 2] and not img[n // 2, n // 2 + r + 1]
    for s in (img[::-1], img[:, ::-1], img.T):
        assert np.array_equal(img, s)


def test_box_blur_equals_scipy_uniform_filter_with_reflect101_border():
    """cv2.blur(mask, (3, 3)) of reveal_fog_of_war's smoothing [frontier_exploration, ext]: the 3 x 3 mean with BORDER_REFLECT_101,
    rounded half to even... or half up?  The stand-in's choice is checked against the exact rational mean: a 3 x 3 sum of u8 values
    divided by 9 is never at a .5 tie (9 is odd), so ANY correct rounding gives round(sum / 9)."""
    rng = np.random.default_rng(0)
    for _ in range(5):
        img = (rng.integers(0, 2, (23, 31)) * 255).astype(np.uint8)
        p = np.pad(img.astype(np.int64), 1, mode="reflect")            # numpy "reflect" = OpenCV REFLECT_101
        s = sum(p[dy:dy + 23, dx:dx + 31] for dy in range(3) for dx in range(3))
        assert np.array_equal(cv.blur(img, (3, 3)), np.rint(s / 9.0).astype(np.uint8))


def test_contour_area_point_test_and_convexity_against_first_principles():
    """contourArea = |shoelace| (Green's formula on the vertices); pointPolygonTest(measureDist=False) = +1 / 0 / -1 for inside / on an
    edge / outside, decided here by exact integer cross products (winding number); isContourConvex = all turns of one sign."""
    rng = np.random.default_rng(3)
    for trial in range(200):
        n = int(rng.integers(3, 9))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = rng.uniform(3, 20, n) if trial % 2 else np.full(n, 15.0)
        pts = np.unique(np.rint(np.stack([25 + rad * np.cos(ang), 25 + rad * np.sin(ang)], 1)).astype(np.int64), axis=0)
        if len(pts) < 3:
            continue
        ctr = pts.mean(0)
        pts = pts[np.argsort(np.arctan2(pts[:, 1] - ctr[1], pts[:, 0] - ctr[0]))]          # a simple (star-shaped) polygon
        x, y = pts[:, 0], pts[:, 1]
        shoelace = abs(int(np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y))) / 2.0
        assert cv.contourArea(pts) == shoelace
        turns = [(int(pts[(i + 1) % len(pts)][0] - pts[i][0]) * int(pts[(i + 2) % len(pts)][1] - pts[(i + 1) % len(pts)][1])
                  - int(pts[(i + 1) % len(pts)][1] - pts[i][1]) * int(pts[(i + 2) % len(pts)][0] - pts[(i + 1) % len(pts)][0]))
                 for i in range(len(pts))]
        strictly = all(t > 0 for t in turns) or all(t < 0 for t in turns)
        if strictly:
            assert cv.isContourConvex(pts)
        elif any(t > 0 for t in turns) and any(t < 0 for t in turns):
            assert not cv.isContourConvex(pts)
        for _ in range(20):
            q = rng.integers(0, 51, 2)
            on_edge, wn = False, 0
            for i in range(len(pts)):
                a, b = pts[i], pts[(i + 1) % len(pts)]
                cr = int(b[0] - a[0]) * int(q[1] - a[1]) - int(b[1] - a[1]) * int(q[0] - a[0])
                if cr == 0 and min(a[0], b[0]) <= q[0] <= max(a[0], b[0]) and min(a[1], b[1]) <= q[1] <= max(a[1], b[1]):
                    on_edge = True
                if a[1] <= q[1] < b[1] and cr > 0:
                    wn += 1
                elif b[1] <= q[1] < a[1] and cr < 0:
                    wn -= 1
            want = 0.0 if on_edge else (1.0 if wn != 0 else -1.0)
            assert cv.pointPolygonTest(pts, (float(q[0]), float(q[1])), False) == want, (pts.tolist(), q.tolist())


def test_bounding_rect_of_a_mask_against_numpy():
    rng = np.random.default_rng(4)
    for _ in range(20):
        m = np.zeros((40, 50), np.uint8)
        y0, x0 = int(rng.integers(0, 30)), int(rng.integers(0, 40))
        m[y0:y0 + int(rng.integers(1, 10)), x0:x0 + int(rng.integers(1, 10))] = 255
        m[rng.integers(0, 40), rng.integers(0, 50)] = 1
        ys, xs = np.nonzero(m)
        assert cv.boundingRect(m) == (xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1)
    assert cv.boundingRect(np.zeros((5, 5), np.uint8)) == (0, 0, 0, 0)


@pytest.mark.parametrize("case", [(100, 100, 60, 0, -40, 40), (50, 70, 25, 90, 10, 170), (100, 100, 100, 37, -39, 39), (30, 30, 7, 0, 0, 360)])
def test_ellipse_polygon_vertices_lie_on_the_analytic_arc(case):
    """cv2.ellipse2Poly behind the cone template (value_map.py:325-334) and the fog-of-war wedge: integer vertices of a circular arc --
    the vertices EllipseEx hands to the fill in 16.16 fixed point -- each within one pixel of the analytic circle (the double-precision
    ellipse2Poly points, rounded to 1/65536),
    angularly ordered from the start to the end angle (image coordinates: y down, angles measured from +x towards +y)."""
    cx, cy, r, rot, a0, a1 = case
    poly = cv.ellipse_polygon((cx, cy), (r, r), rot, a0, a1).reshape(-1, 2).astype(np.float64) / 65536.0     # 16.16 fixed point
    assert len(poly) >= 3
    d = np.hypot(poly[:, 0] - cx, poly[:, 1] - cy)
    on_arc = np.abs(d - r) <= 1.0
    centre = (poly[:, 0] == cx) & (poly[:, 1] == cy)            # a sector polygon may carry the centre as its apex
    assert (on_arc | centre).all(), poly[~(on_arc | centre)]
    arc = poly[on_arc]
    ang = np.unwrap(np.arctan2(arc[:, 1] - cy, arc[:, 0] - cx))
    ang = np.rad2deg(ang) - rot
    ang -= 360.0 * np.round((ang[0] - a0) / 360.0)
    tol = np.rad2deg(1.5 / r)                                    # one pixel of arc length
    assert abs(ang[0] - a0) <= tol and abs(ang[-1] - a1) <= tol, (ang[0], ang[-1])
    assert (np.diff(ang) >= -tol).all()                          # monotone up to the rounding of neighbouring vertices
    if a1 - a0 < 360:
        assert (np.diff(ang) <= 15 + tol).all()                  # no gap larger than the coarsest delta OpenCV uses (it picks 1-15 degrees by size)

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

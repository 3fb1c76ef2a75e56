"""oracle/cv.py -- TEST INFRASTRUCTURE ONLY.

A tiny ``cv2``-shaped facade over ``oracle/cvport.c`` so that the oracle modules
(`oracle/value_map.py`, `oracle/obstacle_map.py`, ...) can follow the reference
source line by line (``cv2.ellipse`` -> ``cv.ellipse`` etc.).  OpenCV itself is not
available in this environment (pinned opencv-python==4.5.5.64,
/root/reference/pyproject.toml:28); parity with real OpenCV is UNPINNED HERE.

On a machine that HAS the pinned wheel, ``VLFM_REAL_CV2=1`` rebinds every facade function below to the real ``cv2``
(``use_real()``), so the same oracle modules and the same tests run on OpenCV itself and any deviation of cvport.c shows up
as a diff against the committed fixtures (tools/verify_with_real_vlfm.md).  ``STANDIN`` keeps the cvport versions reachable
for the side-by-side comparison in tests/test_oracle_cv.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcvport.so")

RETR_EXTERNAL, RETR_LIST, RETR_CCOMP, RETR_TREE = 0, 1, 2, 3
CHAIN_APPROX_NONE, CHAIN_APPROX_SIMPLE = 1, 2


def build(force: bool = False) -> str:
    """Compile oracle/cvport.c with gcc (seconds)."""
    src = os.path.join(_HERE, "cvport.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(
            ["gcc", "-O3", "-fPIC", "-shared", "-std=c99", "-ffp-contract=off", "-o", _SO, src, "-lm"]
        )
    return _SO


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
        i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
        i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
        f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        ci, cd, cl = ctypes.c_int, ctypes.c_double, ctypes.c_long
        L.cvp_line8.argtypes = [u8p, ci, ci, cl, cl, cl, cl, ci]
        L.cvp_fill_poly.argtypes = [u8p, ci, ci, i32p, i32p, ci, ci]
        L.cvp_circle_fill.argtypes = [u8p, ci, ci, ci, ci, ci, ci]
        L.cvp_polylines2_thick.argtypes = [u8p, ci, ci, i32p, ci, ci, ci]
        L.cvp_ellipse_poly.argtypes = [ci, ci, ci, ci, cd, cd, cd, i64p]
        L.cvp_ellipse_poly.restype = ci
        L.cvp_ellipse_fill.argtypes = [u8p, ci, ci, ci, ci, ci, ci, cd, cd, cd, ci]
        L.cvp_warp_affine_f64.argtypes = [f64p, f64p, ci, ci, f64p, cd]
        L.cvp_dilate_rect.argtypes = [u8p, u8p, ci, ci, ci, ci]
        L.cvp_blur3x3.argtypes = [u8p, u8p, ci, ci]
        L.cvp_find_contours.argtypes = [u8p, ci, ci, ci, ci, i32p, cl, i32p, i32p, ci]
        L.cvp_find_contours.restype = ci
        L.cvp_contour_area.argtypes = [i32p, ci]
        L.cvp_contour_area.restype = cd
        L.cvp_point_polygon_test.argtypes = [i32p, ci, cd, cd, ci]
        L.cvp_point_polygon_test.restype = cd
        L.cvp_is_contour_convex.argtypes = [i32p, ci]
        L.cvp_is_contour_convex.restype = ci
        L.cvp_contour_frontier_midpoints.argtypes = [i32p, ci, u8p, ci, ci, f64p, ci]
        L.cvp_contour_frontier_midpoints.restype = ci
        _lib = L
    return _lib


def _cvround(v: float) -> int:
    return int(np.rint(v))  # round-half-even == cvRound (lrint)


def _draw(img: np.ndarray, color, painter) -> np.ndarray:
    """Run ``painter(mask_u8)`` and assign ``color`` where it drew.  Drawing == 'set pixel to color',
    so painting a u8 mask and assigning through it is exact for every dtype (u8, f64 cone mask...)."""
    assert img.ndim == 2
    if img.dtype == np.uint8 and img.flags.c_contiguous:
        painter(img, int(color))
        return img
    mask = np.zeros(img.shape, np.uint8)
    painter(mask, 1)
    img[mask > 0] = color
    return img


def ellipse(img, center, axes, angle, startAngle, endAngle, color, thickness=-1):
    assert thickness < 0, "only filled ellipses are on the path"
    r, c = img.shape
    return _draw(
        img,
        color,
        lambda m, col: lib().cvp_ellipse_fill(
            m, r, c, int(center[0]), int(center[1]), int(axes[0]), int(axes[1]),
            float(angle), float(startAngle), float(endAngle), col),
    )


def ellipse_polygon(center, axes, angle, startAngle, endAngle) -> np.ndarray:
    """The 16.16 fixed-point polygon EllipseEx rasterises (incl. the appended centre for sectors)."""
    out = np.zeros(2 * 404, np.int64)
    n = lib().cvp_ellipse_poly(int(center[0]), int(center[1]), int(axes[0]), int(axes[1]),
                               float(angle), float(startAngle), float(endAngle), out)
    return out[: 2 * n].reshape(n, 2).copy()


def _pack_contours(contours: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    pts = [np.asarray(c).reshape(-1, 2).astype(np.int32) for c in contours]
    lens = np.array([len(p) for p in pts], np.int32)
    flat = np.ascontiguousarray(np.concatenate(pts, axis=0) if pts else np.zeros((0, 2), np.int32))
    if flat.size == 0:
        flat = np.zeros((1, 2), np.int32)
    return flat, lens


def drawContours(img, contours, contourIdx, color, thickness=-1):
    assert thickness < 0, "only filled contours are on the path"
    sel = list(contours) if contourIdx < 0 else [contours[contourIdx]]
    if len(sel) == 0:
        return img
    flat, lens = _pack_contours(sel)
    r, c = img.shape
    return _draw(img, color, lambda m, col: lib().cvp_fill_poly(m, r, c, flat, lens, len(lens), col))


def circle(img, center, radius, color, thickness=-1):
    assert thickness < 0
    r, c = img.shape
    return _draw(img, color,
                 lambda m, col: lib().cvp_circle_fill(m, r, c, int(center[0]), int(center[1]), int(radius), col))


def polylines(img, pts, isClosed, color, thickness=1):
    """Only the form frontier_exploration uses: (N,2,2) int32 open two-point lines, thickness 2."""
    assert not isClosed and thickness > 1
    segs = np.ascontiguousarray(np.asarray(pts, np.int32).reshape(-1, 4))
    r, c = img.shape
    return _draw(img, color,
                 lambda m, col: lib().cvp_polylines2_thick(m, r, c, segs if len(segs) else np.zeros((1, 4), np.int32),
                                                           len(segs), col, int(thickness)))


def getRotationMatrix2D(center, angle, scale):
    import math

    angle = float(angle) * (np.pi / 180)  # cv: angle *= CV_PI/180 (constant folded first)
    # cv::getRotationMatrix2D is C++: std::cos/std::sin = libm (NumPy's SIMD cos/sin differ from libm by a few ulp)
    alpha = math.cos(angle) * scale
    beta = math.sin(angle) * scale
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))  # Point2f
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy]])


def warpAffine(src, M, dsize, borderValue=0):
    assert src.dtype == np.float64 and src.ndim == 2 and dsize == (src.shape[1], src.shape[0])
    src = np.ascontiguousarray(src)
    dst = np.empty_like(src)
    lib().cvp_warp_affine_f64(src, dst, src.shape[0], src.shape[1], np.ascontiguousarray(M, np.float64).reshape(-1),
                              float(borderValue))
    return dst


def dilate(src, kernel, iterations=1):
    assert iterations == 1 and src.dtype == np.uint8 and np.all(kernel == 1)
    src = np.ascontiguousarray(src)
    dst = np.empty_like(src)
    lib().cvp_dilate_rect(src, dst, src.shape[0], src.shape[1], kernel.shape[1], kernel.shape[0])
    return dst


def blur(src, ksize):
    assert tuple(ksize) == (3, 3) and src.dtype == np.uint8
    src = np.ascontiguousarray(src)
    dst = np.empty_like(src)
    lib().cvp_blur3x3(src, dst, src.shape[0], src.shape[1])
    return dst


def findContours(image, mode, method) -> Tuple[List[np.ndarray], None]:
    """Returns (contours, None); contours are (n,1,2) int32 in OpenCV's order.  RETR_CCOMP/RETR_TREE are traced
    as RETR_LIST (same borders; the hierarchy is not produced -- no caller on the path reads it)."""
    img = np.ascontiguousarray(image, np.uint8)
    rows, cols = img.shape
    m = 0 if mode == RETR_EXTERNAL else 1
    cap_pts, cap_c = max(4 * (rows + cols), 1 << 16), 4096
    while True:
        pts = np.zeros((cap_pts, 2), np.int32)
        lens = np.zeros(cap_c, np.int32)
        holes = np.zeros(cap_c, np.int32)
        n = lib().cvp_find_contours(img, rows, cols, m, method, pts.reshape(-1), cap_pts, lens, holes, cap_c)
        if n >= 0:
            break
        cap_pts *= 4
        cap_c *= 4
    out, o = [], 0
    for k in range(n):
        out.append(pts[o:o + lens[k]].reshape(-1, 1, 2).copy())
        o += lens[k]
    return out, None


def contourArea(cnt) -> float:
    p = np.ascontiguousarray(np.asarray(cnt).reshape(-1, 2), np.int32)
    return float(lib().cvp_contour_area(p if len(p) else np.zeros((1, 2), np.int32), len(p)))


def pointPolygonTest(cnt, pt, measureDist) -> float:
    p = np.ascontiguousarray(np.asarray(cnt).reshape(-1, 2), np.int32)
    return float(lib().cvp_point_polygon_test(p, len(p), float(pt[0]), float(pt[1]), int(bool(measureDist))))


def isContourConvex(cnt) -> bool:
    p = np.ascontiguousarray(np.asarray(cnt).reshape(-1, 2), np.int32)
    return bool(lib().cvp_is_contour_convex(p, len(p)))


def bitwise_and(a, b):
    return np.bitwise_and(a, b)


def erode(src, kernel=None, iterations=1):
    """cv2.erode with the default 3x3 rectangular element (kernel=None), border = morphologyDefaultBorderValue."""
    assert kernel is None
    from .ref_object_map import erode3x3

    out = erode3x3(src, iterations)
    return np.where(out > 0, np.asarray(src).max() if np.asarray(src).size else 0, 0).astype(np.asarray(src).dtype)


def boundingRect(array):
    """cv2.boundingRect of a mask image: bounding box of the non-zero pixels (x, y, w, h); all zeros -> (0, 0, 0, 0)."""
    a = np.asarray(array)
    ys, xs = np.nonzero(a)
    if len(xs) == 0:
        return (0, 0, 0, 0)
    return (int(xs.min()), int(ys.min()), int(xs.max() - xs.min() + 1), int(ys.max() - ys.min() + 1))


# ---------------------------------------------------------------------------------------------- real-library switch
FACADE_NAMES = ("ellipse", "drawContours", "circle", "polylines", "getRotationMatrix2D", "warpAffine", "dilate", "blur",
                "findContours", "contourArea", "pointPolygonTest", "isContourConvex", "bitwise_and", "erode", "boundingRect")
STANDIN = {n: globals()[n] for n in FACADE_NAMES}      # the cvport.c versions, whatever the module-level names point at
BACKEND = "stand-in: oracle/cvport.c (OpenCV 4.5.5 restated)"


def real_cv2_requested() -> bool:
    return os.environ.get("VLFM_REAL_CV2", "") not in ("", "0")


def import_real_cv2():
    """The real ``cv2`` package, or ImportError: a module named cv2 that carries ref_shim's stand-in mark does not count."""
    import importlib
    import sys

    mod = sys.modules.get("cv2")
    if mod is not None and getattr(mod, "__vlfm_standin__", False):
        del sys.modules["cv2"]
        mod = None
    mod = mod or importlib.import_module("cv2")
    if getattr(mod, "__vlfm_standin__", False):
        raise ImportError("the module named cv2 is oracle/ref_shim.py's stand-in")
    return mod


def use_real(cv2_module=None) -> str:
    """Rebind the facade to a real OpenCV: every oracle module that says ``cv.ellipse`` / ``cv.warpAffine`` ... now runs
    OpenCV's own code.  Returns the backend description."""
    global BACKEND
    real = cv2_module or import_real_cv2()
    g = globals()
    for n in FACADE_NAMES:
        g[n] = getattr(real, n)
    for n in ("RETR_EXTERNAL", "RETR_LIST", "RETR_CCOMP", "RETR_TREE", "CHAIN_APPROX_NONE", "CHAIN_APPROX_SIMPLE"):
        g[n] = getattr(real, n)
    _find = real.findContours

    def findContours(image, mode, method):   # OpenCV 3.x returned (image, contours, hierarchy); 4.x (contours, hierarchy)
        out = _find(image, mode, method)
        return (list(out[-2]), out[-1])

    g["findContours"] = findContours
    BACKEND = f"real: cv2 {getattr(real, '__version__', '?')} ({getattr(real, '__file__', '?')})"
    return BACKEND


def use_standin() -> None:
    global BACKEND
    globals().update(STANDIN)
    globals().update(RETR_EXTERNAL=0, RETR_LIST=1, RETR_CCOMP=2, RETR_TREE=3, CHAIN_APPROX_NONE=1, CHAIN_APPROX_SIMPLE=2)
    BACKEND = "stand-in: oracle/cvport.c (OpenCV 4.5.5 restated)"


if real_cv2_requested():
    use_real()   # ImportError here is the loud failure VLFM_REAL_CV2=1 asks for

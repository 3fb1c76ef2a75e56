"""oracle/ref_geometry.py -- TEST INFRASTRUCTURE ONLY (parity oracle).

NumPy restatement of the pose / point-cloud / image helpers the mapping path calls.
Each function cites the reference lines whose arithmetic (dtype, op order, rounding) it follows.
PARITY UNPINNED for the OpenCV-backed helpers (see oracle/cvport.c header).
"""
from __future__ import annotations

import math

import numpy as np

from . import cv


def yaw_of(tf: np.ndarray) -> float:
    """vlfm/utils/geometry_utils.py:145-159 (extract_yaw): atan2(R10, R00)."""
    assert tf.shape == (4, 4), "The input matrix must be 4x4"
    return np.arctan2(tf[1, 0], tf[0, 0])


def pose_to_tf(xyz, yaw: float) -> np.ndarray:
    """geometry_utils.py:162-180 (xyz_yaw_to_tf_matrix)."""
    x, y, z = xyz
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0, x], [s, c, 0, y], [0, 0, 1, z], [0, 0, 0, 1]])


def fov_from_focal(focal_length: float, extent: int) -> float:
    """geometry_utils.py:239-254 (get_fov)."""
    return 2 * math.atan((extent / 2) / focal_length)


def unproject(depth_m: np.ndarray, mask: np.ndarray, fx: float, fy: float) -> np.ndarray:
    """geometry_utils.py:216-236 (get_point_cloud): rows of (z, -x, -y); int64*f32 -> f64, /fx."""
    v, u = np.where(mask)
    z = depth_m[v, u]
    x = (u - depth_m.shape[1] // 2) * z / fx
    y = (v - depth_m.shape[0] // 2) * z / fy
    return np.stack((z, -x, -y), axis=-1)


def apply_tf(tf: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """geometry_utils.py:205-213 (transform_points): homogeneous multiply then divide by w."""
    hom = np.hstack((pts, np.ones((pts.shape[0], 1))))
    out = np.dot(tf, hom.T).T
    return out[:, :3] / out[:, 3:]


def rotate_about_centre(image: np.ndarray, radians: float, border_value=0) -> np.ndarray:
    """vlfm/utils/img_utils.py:9-28 (rotate_image): getRotationMatrix2D((w//2,h//2), degrees) + warpAffine."""
    h, w = image.shape[0], image.shape[1]
    M = cv.getRotationMatrix2D((w // 2, h // 2), np.degrees(radians), 1.0)
    return cv.warpAffine(image, M, (w, h), borderValue=border_value)


def paste_centred(base: np.ndarray, patch: np.ndarray, row: int, col: int) -> np.ndarray:
    """img_utils.py:31-61 (place_img_in_img): patch centre lands on (row, col); both sides clipped."""
    assert 0 <= row < base.shape[0] and 0 <= col < base.shape[1], "Pixel location is outside the image."
    top, left = row - patch.shape[0] // 2, col - patch.shape[1] // 2
    bottom, right = top + patch.shape[0], left + patch.shape[1]
    bt, bl = max(0, top), max(0, left)
    bb, br = min(base.shape[0], bottom), min(base.shape[1], right)
    pt, pl = max(0, -top), max(0, -left)
    base[bt:bb, bl:br] = patch[pt:pt + (bb - bt), pl:pl + (br - bl)]
    return base


def disc_reduce(image: np.ndarray, pixel_location, radius: int, reduction: str = "median"):
    """img_utils.py:213-266 (pixel_value_within_radius).  Note the disc is drawn at (radius, radius) of the
    *clipped* crop (img_utils.py:246-253), zeros are excluded, and an empty selection yields -1."""
    assert (0 <= pixel_location[0] < image.shape[0] and 0 <= pixel_location[1] < image.shape[1]), \
        "Pixel location is outside the image."
    r0 = max(0, pixel_location[0] - radius)
    c0 = max(0, pixel_location[1] - radius)
    r1 = min(image.shape[0], pixel_location[0] + radius + 1)
    c1 = min(image.shape[1], pixel_location[1] + radius + 1)
    crop = image[r0:r1, c0:c1]
    disc = np.zeros(crop.shape[:2], dtype=np.uint8)
    disc = cv.circle(disc, (radius, radius), radius, color=255, thickness=-1)
    vals = crop[disc > 0]
    vals = vals[vals > 0]
    if vals.size == 0:
        return -1
    if reduction == "mean":
        return np.mean(vals)
    if reduction == "max":
        return np.max(vals)
    if reduction == "median":
        return np.median(vals)
    raise ValueError(f"Invalid reduction method: {reduction}")


def fill_small_holes(depth_img: np.ndarray, area_thresh: int) -> np.ndarray:
    """img_utils.py:361-390: zero-depth regions whose contour area < thresh become 1.0.
    (RETR_TREE there; every border is traced either way and the hierarchy is discarded.)"""
    holes = np.where(depth_img == 0, 1, 0).astype("uint8")
    contours, _ = cv.findContours(holes, cv.RETR_TREE, cv.CHAIN_APPROX_SIMPLE)
    filled = np.zeros_like(holes)
    for cnt in contours:
        if cv.contourArea(cnt) < area_thresh:
            cv.drawContours(filled, [cnt], 0, 1, -1)
    return np.where(filled == 1, 1, depth_img)

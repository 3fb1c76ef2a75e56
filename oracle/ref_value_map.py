"""oracle/ref_value_map.py -- TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline).

NumPy restatement of the reference's ValueMap update path, following the arithmetic of
/root/reference/vlfm/mapping/base_map.py and /root/reference/vlfm/mapping/value_map.py
expression by expression (dtype promotions and truncations included -- SURVEY.md App. A).

PARITY UNPINNED: the reference ships no golden vectors for this path and its OpenCV calls go
through oracle/cvport.c (a restatement of OpenCV 4.5.5, not OpenCV).  Pins that exist are the
hand-derived known answers in tests/test_oracle_value_map.py.
"""
from __future__ import annotations

import os
import warnings
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np

from . import cv
from .ref_geometry import disc_reduce, paste_centred, rotate_about_centre, yaw_of


class RefBaseMap:
    """base_map.py:10-60."""

    _map_dtype = np.dtype(np.float32)

    def __init__(self, size: int = 1000, pixels_per_meter: int = 20):
        self.pixels_per_meter = pixels_per_meter
        self.size = size
        self._map = np.zeros((size, size), dtype=self._map_dtype)
        self._episode_pixel_origin = np.array([size // 2, size // 2])
        self._camera_positions: List[np.ndarray] = []
        self._last_camera_yaw = 0.0

    def reset(self) -> None:
        self._map.fill(0)
        self._camera_positions = []

    def update_agent_traj(self, robot_xy, robot_heading) -> None:
        self._camera_positions.append(robot_xy)
        self._last_camera_yaw = robot_heading

    def _xy_to_px(self, points: np.ndarray) -> np.ndarray:
        # base_map.py:44-46 -- rint (half-even) of (y,x)*ppm, + origin, first column flipped about the map height
        px = np.rint(points[:, ::-1] * self.pixels_per_meter) + self._episode_pixel_origin
        px[:, 0] = self._map.shape[0] - px[:, 0]
        return px.astype(int)

    def _px_to_xy(self, px: np.ndarray) -> np.ndarray:
        # base_map.py:57-60
        q = px.copy()
        q[:, 0] = self._map.shape[0] - q[:, 0]
        pts = (q - self._episode_pixel_origin) / self.pixels_per_meter
        return pts[:, ::-1]


def _remap(value, from_low, from_high, to_low, to_high):
    """value_map.py:432-445."""
    return (value - from_low) * (to_high - to_low) / (from_high - from_low) + to_low


class RefValueMap(RefBaseMap):
    """value_map.py:33-429 (without recording / visualisation)."""

    _confidence_masks: Dict[Tuple[float, float], np.ndarray] = {}
    _min_confidence = 0.25
    _decision_threshold = 0.35

    def __init__(self, value_channels: int, size: int = 1000, use_max_confidence: bool = True,
                 fusion_type: str = "default", obstacle_map=None) -> None:
        super().__init__(size)
        self._value_map = np.zeros((size, size, value_channels), np.float32)
        self._value_channels = value_channels
        self._use_max_confidence = use_max_confidence
        self._fusion_type = fusion_type
        self._obstacle_map = obstacle_map
        if obstacle_map is not None:
            assert obstacle_map.pixels_per_meter == self.pixels_per_meter
            assert obstacle_map.size == self.size
        if os.environ.get("MAP_FUSION_TYPE", "") != "":
            self._fusion_type = os.environ["MAP_FUSION_TYPE"]

    def reset(self) -> None:
        super().reset()
        self._value_map.fill(0)

    # ------------------------------------------------------------------ value_map.py:100-128
    def update_map(self, values, depth, tf_camera_to_episodic, min_depth, max_depth, fov) -> None:
        assert len(values) == self._value_channels, \
            f"Incorrect number of values given ({len(values)}). Expected {self._value_channels}."
        curr_map = self._localize_new_data(depth, tf_camera_to_episodic, min_depth, max_depth, fov)
        self._fuse_new_data(curr_map, values)

    # ------------------------------------------------------------------ value_map.py:146-187
    def sort_waypoints(self, waypoints, radius: float, reduce_fn: Optional[Callable] = None):
        radius_px = int(radius * self.pixels_per_meter)

        def score(point):
            x, y = point
            px = int(-x * self.pixels_per_meter) + self._episode_pixel_origin[0]
            py = int(-y * self.pixels_per_meter) + self._episode_pixel_origin[1]
            cell = (self._value_map.shape[0] - px, py)
            per_ch = [disc_reduce(self._value_map[..., c], cell, radius_px) for c in range(self._value_channels)]
            return per_ch[0] if len(per_ch) == 1 else tuple(per_ch)

        values = [score(p) for p in waypoints]
        if self._value_channels > 1:
            assert reduce_fn is not None, "Must provide a reduction function when using multiple value channels."
            values = reduce_fn(values)
        order = np.argsort([-v for v in values])
        return np.array([waypoints[i] for i in order]), [values[i] for i in order]

    # ------------------------------------------------------------------ value_map.py:221-260
    def depth_profile_contour(self, depth, fov, min_depth, max_depth, template_shape) -> np.ndarray:
        """The polygon that value_map.py:234-257 hands to drawContours, as (N,2) int (x=col, y=row)."""
        if len(depth.shape) == 3:
            depth = depth.squeeze(2)
        depth_row = np.max(depth, axis=0) * (max_depth - min_depth) + min_depth  # f32
        angles = np.linspace(-fov / 2, fov / 2, len(depth_row))  # f64
        x = depth_row
        y = depth_row * np.tan(angles)  # f64
        x = (x * self.pixels_per_meter + template_shape[0] / 2).astype(int)  # f32 -> trunc
        y = (y * self.pixels_per_meter + template_shape[1] / 2).astype(int)  # f64 -> trunc
        last_row, last_col = template_shape[0] - 1, template_shape[1] - 1
        start = np.array([[0, last_col]])
        end = np.array([[last_row, last_col]])
        return np.concatenate((start, np.stack((y, x), axis=1), end), axis=0)

    def _process_local_data(self, depth, fov, min_depth, max_depth) -> np.ndarray:
        cone_mask = self._get_confidence_mask(fov, max_depth)
        contour = self.depth_profile_contour(depth, fov, min_depth, max_depth, cone_mask.shape)
        return cv.drawContours(cone_mask, [contour], -1, 0, -1)

    # ------------------------------------------------------------------ value_map.py:288-319
    def _localize_new_data(self, depth, tf, min_depth, max_depth, fov) -> np.ndarray:
        curr = self._process_local_data(depth, fov, min_depth, max_depth)
        yaw = yaw_of(tf)
        curr = rotate_about_centre(curr, -yaw)
        cam_x, cam_y = tf[:2, 3] / tf[3, 3]
        px = int(cam_x * self.pixels_per_meter) + self._episode_pixel_origin[0]  # truncation toward 0
        py = int(-cam_y * self.pixels_per_meter) + self._episode_pixel_origin[1]
        curr_map = np.zeros_like(self._map)
        return paste_centred(curr_map, curr, px, py)  # f64 -> f32 on assignment

    # ------------------------------------------------------------------ value_map.py:321-355
    def _get_blank_cone_mask(self, fov, max_depth) -> np.ndarray:
        size = int(max_depth * self.pixels_per_meter)
        cone = np.zeros((size * 2 + 1, size * 2 + 1))
        return cv.ellipse(cone, (size, size), (size, size), 0,
                          -np.rad2deg(fov) / 2 + 90, np.rad2deg(fov) / 2 + 90, 1, -1)

    def _get_confidence_mask(self, fov, max_depth) -> np.ndarray:
        key = (fov, max_depth)
        if key in self._confidence_masks:
            return self._confidence_masks[key].copy()
        cone = self._get_blank_cone_mask(fov, max_depth)
        n_r, n_c = cone.shape
        # value_map.py:343-351: per-cell NumPy *scalar* math (libm atan2/cos/pow, not the SIMD array loops),
        # stored into an f32 array.  Kept as a scalar loop so the last-ulp behaviour is the reference's.
        conf32 = np.zeros((n_r, n_c), np.float32)
        for r in range(n_r):
            off_fwd = abs(r - n_r // 2)
            for c in range(n_c):
                off_lat = abs(c - n_c // 2)
                theta = _remap(np.arctan2(off_lat, off_fwd), 0, fov / 2, 0, np.pi / 2)
                conf32[r, c] = _remap(np.cos(theta) ** 2, 0, 1, self._min_confidence, 1)
        adjusted = conf32 * cone  # f32 * f64 -> f64
        self._confidence_masks[key] = adjusted.copy()
        return adjusted

    # ------------------------------------------------------------------ value_map.py:357-429
    def _fuse_new_data(self, new_map: np.ndarray, values: np.ndarray) -> None:
        assert len(values) == self._value_channels, \
            f"Incorrect number of values given ({len(values)}). Expected {self._value_channels}."
        if self._obstacle_map is not None:
            explored = self._obstacle_map.explored_area
            new_map[explored == 0] = 0
            self._map[explored == 0] = 0
            self._value_map[explored == 0] *= 0

        if self._fusion_type == "replace":
            fresh = np.zeros_like(self._value_map)
            fresh[new_map > 0] = values
            self._map[new_map > 0] = new_map[new_map > 0]
            self._value_map[new_map > 0] = fresh[new_map > 0]
            return
        elif self._fusion_type == "equal_weighting":
            self._map[self._map > 0] = 1
            new_map[new_map > 0] = 1
        else:
            assert self._fusion_type == "default", f"Unknown fusion type {self._fusion_type}"

        silenced = np.logical_and(new_map < self._decision_threshold, new_map < self._map)
        new_map[silenced] = 0

        if self._use_max_confidence:
            higher = new_map > self._map
            self._value_map[higher] = values
            self._map[higher] = new_map[higher]
        else:
            denom = self._map + new_map
            with warnings.catch_warnings():
                warnings.filterwarnings("ignore", category=RuntimeWarning)
                w1 = self._map / denom
                w2 = new_map / denom
            w1c = np.repeat(np.expand_dims(w1, axis=2), self._value_channels, axis=2)
            w2c = np.repeat(np.expand_dims(w2, axis=2), self._value_channels, axis=2)
            self._value_map = self._value_map * w1c + values * w2c  # f64 from here on (values is f64)
            self._map = self._map * w1 + new_map * w2
            self._value_map = np.nan_to_num(self._value_map)
            self._map = np.nan_to_num(self._map)

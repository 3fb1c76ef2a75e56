"""oracle/ref_obstacle_map.py -- TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline).

NumPy restatement of /root/reference/vlfm/mapping/obstacle_map.py:15-169,196-197 (visualisation omitted), with the
OpenCV calls routed through oracle/cv.py and the frontier_exploration calls through
oracle/ref_frontier_exploration.py.  PARITY UNPINNED (see those modules' headers)."""
from __future__ import annotations

import numpy as np

from . import cv
from .ref_frontier_exploration import detect_frontier_waypoints, reveal_fog_of_war
from .ref_geometry import apply_tf, fill_small_holes, unproject, yaw_of
from .ref_value_map import RefBaseMap


def keep_height_band(points: np.ndarray, lo: float, hi: float) -> np.ndarray:
    """obstacle_map.py:196-197."""
    return points[(points[:, 2] >= lo) & (points[:, 2] <= hi)]


class RefObstacleMap(RefBaseMap):
    _map_dtype = np.dtype(bool)

    def __init__(self, min_height: float, max_height: float, agent_radius: float, area_thresh: float = 3.0,
                 hole_area_thresh: int = 100000, size: int = 1000, pixels_per_meter: int = 20):
        super().__init__(size, pixels_per_meter)
        self.explored_area = np.zeros((size, size), dtype=bool)
        self._map = np.zeros((size, size), dtype=bool)
        self._navigable_map = np.zeros((size, size), dtype=bool)
        self._min_height, self._max_height = min_height, max_height
        self._area_thresh_in_pixels = area_thresh * (self.pixels_per_meter ** 2)
        self._hole_area_thresh = hole_area_thresh
        k = self.pixels_per_meter * agent_radius * 2
        k = int(k) + (int(k) % 2 == 0)  # obstacle_map.py:43-45: make it odd
        self._navigable_kernel = np.ones((k, k), np.uint8)
        self._frontiers_px = np.array([])
        self.frontiers = np.array([])

    def reset(self) -> None:
        super().reset()
        self._navigable_map.fill(0)
        self.explored_area.fill(0)
        self._frontiers_px = np.array([])
        self.frontiers = np.array([])

    def update_map(self, depth, tf_camera_to_episodic, min_depth, max_depth, fx, fy, topdown_fov, explore=True,
                   update_obstacles=True) -> None:
        if update_obstacles:  # obstacle_map.py:86-109
            if self._hole_area_thresh == -1:
                filled = depth.copy()
                filled[depth == 0] = 1.0
            else:
                filled = fill_small_holes(depth, self._hole_area_thresh)
            scaled = filled * (max_depth - min_depth) + min_depth
            mask = scaled < max_depth
            cloud_cam = unproject(scaled, mask, fx, fy)
            cloud_epi = apply_tf(tf_camera_to_episodic, cloud_cam)
            obstacles = keep_height_band(cloud_epi, self._min_height, self._max_height)
            px = self._xy_to_px(obstacles[:, :2])
            self._map[px[:, 1], px[:, 0]] = 1
            self._navigable_map = 1 - cv.dilate(self._map.astype(np.uint8), self._navigable_kernel,
                                                iterations=1).astype(bool)
        if not explore:
            return
        # obstacle_map.py:115-153
        agent_xy = tf_camera_to_episodic[:2, 3]
        agent_px = self._xy_to_px(agent_xy.reshape(1, 2))[0]
        new_explored = reveal_fog_of_war(
            top_down_map=self._navigable_map.astype(np.uint8),
            current_fog_of_war_mask=np.zeros_like(self._map, dtype=np.uint8),
            current_point=agent_px[::-1],
            current_angle=-yaw_of(tf_camera_to_episodic),
            fov=np.rad2deg(topdown_fov),
            max_line_len=max_depth * self.pixels_per_meter,
        )
        new_explored = cv.dilate(new_explored, np.ones((3, 3), np.uint8), iterations=1)
        self.explored_area[new_explored > 0] = 1
        self.explored_area[self._navigable_map == 0] = 0
        contours, _ = cv.findContours(self.explored_area.astype(np.uint8), cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
        if len(contours) > 1:
            min_dist, best_idx = np.inf, 0
            for idx, cnt in enumerate(contours):
                dist = cv.pointPolygonTest(cnt, tuple([int(i) for i in agent_px]), True)
                if dist >= 0:
                    best_idx = idx
                    break
                elif abs(dist) < min_dist:
                    min_dist = abs(dist)
                    best_idx = idx
            new_area = np.zeros_like(self.explored_area, dtype=np.uint8)
            cv.drawContours(new_area, contours, best_idx, 1, -1)
            self.explored_area = new_area.astype(bool)
        self._frontiers_px = self._get_frontiers()
        if len(self._frontiers_px) == 0:
            self.frontiers = np.array([])
        else:
            self.frontiers = self._px_to_xy(self._frontiers_px)

    def _get_frontiers(self) -> np.ndarray:
        """obstacle_map.py:155-169."""
        explored = cv.dilate(self.explored_area.astype(np.uint8), np.ones((5, 5), np.uint8), iterations=1)
        return detect_frontier_waypoints(self._navigable_map.astype(np.uint8), explored, self._area_thresh_in_pixels)

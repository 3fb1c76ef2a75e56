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

    # ------------------------------------------------------------------------------------------ obstacle_map.py:86-109
    def _scatter_obstacles(self, depth, tf, min_depth, max_depth, fx, fy) -> None:
        """Depth holes -> metres -> camera-frame cloud of everything nearer than max_depth -> episodic frame -> height
        band -> cells set; the navigable map is the complement of the obstacles grown by the robot's footprint."""
        if self._hole_area_thresh == -1:
            patched = np.where(depth == 0, depth.dtype.type(1.0), depth)
        else:
            patched = fill_small_holes(depth, self._hole_area_thresh)
        metres = patched * (max_depth - min_depth) + min_depth
        world = apply_tf(tf, unproject(metres, metres < max_depth, fx, fy))
        cells = self._xy_to_px(keep_height_band(world, self._min_height, self._max_height)[:, :2])
        self._map[cells[:, 1], cells[:, 0]] = 1          # NumPy semantics on purpose: IndexError / negative wrap
        grown = cv.dilate(self._map.astype(np.uint8), self._navigable_kernel, iterations=1)
        self._navigable_map = 1 - grown.astype(bool)     # int64 0/1 array, like the reference's

    # ------------------------------------------------------------------------------------------ obstacle_map.py:115-153
    def _reveal(self, tf, max_depth, topdown_fov, agent_cell) -> None:
        seen = reveal_fog_of_war(top_down_map=self._navigable_map.astype(np.uint8),
                                 current_fog_of_war_mask=np.zeros_like(self._map, dtype=np.uint8),
                                 current_point=agent_cell[::-1], current_angle=-yaw_of(tf),
                                 fov=np.rad2deg(topdown_fov), max_line_len=max_depth * self.pixels_per_meter)
        seen = cv.dilate(seen, np.ones((3, 3), np.uint8), iterations=1)
        self.explored_area[seen > 0] = 1
        self.explored_area[self._navigable_map == 0] = 0

    def _keep_agent_component(self, agent_cell) -> None:
        """Several disjoint explored regions: keep the one that contains the agent, else the nearest one (first wins)."""
        regions, _ = cv.findContours(self.explored_area.astype(np.uint8), cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
        if len(regions) <= 1:
            return
        here = tuple(int(v) for v in agent_cell)
        chosen, nearest = 0, np.inf
        for k, region in enumerate(regions):
            signed = cv.pointPolygonTest(region, here, True)
            if signed >= 0:
                chosen = k
                break
            if abs(signed) < nearest:
                chosen, nearest = k, abs(signed)
        canvas = np.zeros_like(self.explored_area, dtype=np.uint8)
        cv.drawContours(canvas, regions, chosen, 1, -1)
        self.explored_area = canvas.astype(bool)

    def update_map(self, depth, tf_camera_to_episodic, min_depth, max_depth, fx, fy, topdown_fov, explore=True,
                   update_obstacles=True) -> None:
        if update_obstacles:
            self._scatter_obstacles(depth, tf_camera_to_episodic, min_depth, max_depth, fx, fy)
        if not explore:
            return
        agent_cell = self._xy_to_px(tf_camera_to_episodic[:2, 3].reshape(1, 2))[0]
        self._reveal(tf_camera_to_episodic, max_depth, topdown_fov, agent_cell)
        self._keep_agent_component(agent_cell)
        self._frontiers_px = self._get_frontiers()
        self.frontiers = self._px_to_xy(self._frontiers_px) if len(self._frontiers_px) else np.array([])

    def _get_frontiers(self) -> np.ndarray:
        """obstacle_map.py:155-169."""
        grown = cv.dilate(self.explored_area.astype(np.uint8), np.ones((5, 5), np.uint8), iterations=1)
        return detect_frontier_waypoints(self._navigable_map.astype(np.uint8), grown, self._area_thresh_in_pixels)

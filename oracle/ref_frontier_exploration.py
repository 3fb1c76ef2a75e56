"""oracle/ref_frontier_exploration.py -- TEST INFRASTRUCTURE ONLY (parity oracle).

Restatement BY CONTRACT of the two third-party routines ObstacleMap calls
(/root/reference/vlfm/mapping/obstacle_map.py:7-8,117-124,164-168):

    frontier_exploration.utils.fog_of_war.reveal_fog_of_war
    frontier_exploration.frontier_detection.detect_frontier_waypoints

The package (github.com/naokiyokoyama/frontier_exploration, un-pinned git HEAD in /root/reference/pyproject.toml:25)
is NOT in /root/reference and not installable here.  What follows restates its published algorithm as recalled
(SURVEY.md Appendix B2/B3); PARITY UNPINNED -- there is no way to check it against the real package in this
environment, and the reference holds no golden vectors for it.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from . import cv


def wrap_heading(theta):
    return (theta + np.pi) % (2 * np.pi) - np.pi


# ------------------------------------------------------------------------------------------------ fog of war (B2)
def get_two_farthest_points(source, cnt, agent_angle):
    """The two contour points subtending the smallest / largest angle at `source` (upstream passes the agent angle in
    DEGREES to a rotation built with cos/sin of radians; kept as is)."""
    pts = cnt.reshape(-1, 2)
    pts = pts - source
    rotation_matrix = np.array([[np.cos(-agent_angle), -np.sin(-agent_angle)],
                                [np.sin(-agent_angle), np.cos(-agent_angle)]])
    pts = np.matmul(pts, rotation_matrix)
    angles = np.arctan2(pts[:, 1], pts[:, 0])
    return cnt[np.argmin(angles)], cnt[np.argmax(angles)]


def vectorize_get_line_points(current_point, points, max_line_len):
    angles = np.arctan2(points[..., 1] - current_point[1], points[..., 0] - current_point[0])
    endpoints = np.stack((points[..., 0] + max_line_len * np.cos(angles),
                          points[..., 1] + max_line_len * np.sin(angles)), axis=-1)
    endpoints = endpoints.astype(np.int32)
    return np.stack([points.reshape(-1, 2), endpoints.reshape(-1, 2)], axis=1)


def reveal_fog_of_war(top_down_map: np.ndarray, current_fog_of_war_mask: np.ndarray, current_point: np.ndarray,
                      current_angle: float, fov: float = 90, max_line_len: float = 100) -> np.ndarray:
    curr_pt_cv2 = current_point[::-1].astype(int)
    angle_cv2 = np.rad2deg(wrap_heading(-current_angle + np.pi / 2))
    cone_mask = cv.ellipse(np.zeros_like(top_down_map), curr_pt_cv2, (int(max_line_len), int(max_line_len)), 0,
                           angle_cv2 - fov / 2, angle_cv2 + fov / 2, 1, -1)
    # pixels in the cone that are NOT navigable
    obstacles_in_cone = cv.bitwise_and(cone_mask, 1 - top_down_map)
    obstacle_contours, _ = cv.findContours(obstacles_in_cone, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    if len(obstacle_contours) == 0:
        return current_fog_of_war_mask  # no obstacles in the cone: returned unchanged
    points = []
    for cnt in obstacle_contours:
        if cv.isContourConvex(cnt):
            pt1, pt2 = get_two_farthest_points(curr_pt_cv2, cnt, angle_cv2)
            points.append(pt1.reshape(-1, 2))
            points.append(pt2.reshape(-1, 2))
        else:
            points.append(cnt.reshape(-1, 2))
    points = np.concatenate(points, axis=0)
    visible_cone_mask = cv.bitwise_and(cone_mask, top_down_map)
    line_points = vectorize_get_line_points(curr_pt_cv2, points, max_line_len * 1.05)
    cv.polylines(visible_cone_mask, line_points, isClosed=False, color=0, thickness=2)
    final_contours, _ = cv.findContours(visible_cone_mask, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    visible_area = None
    min_dist = np.inf
    for cnt in final_contours:
        pt = tuple([int(i) for i in curr_pt_cv2])
        dist = abs(cv.pointPolygonTest(cnt, pt, True))
        if dist < min_dist:
            min_dist = dist
            visible_area = cnt
    if min_dist > 3:
        return current_fog_of_war_mask  # the closest visible region is too far from the agent
    return cv.drawContours(current_fog_of_war_mask, [visible_area], 0, 1, -1)


# ------------------------------------------------------------------------------------------------ frontiers (B3)
def filter_out_small_unexplored(full_map: np.ndarray, explored_mask: np.ndarray, area_thresh: int) -> np.ndarray:
    if area_thresh == -1:
        return explored_mask
    unexplored_mask = full_map.copy()
    unexplored_mask[explored_mask > 0] = 0
    contours, _ = cv.findContours(unexplored_mask, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)
    small_contours = []
    for contour in contours:
        if cv.contourArea(contour) < area_thresh:
            mask = np.zeros_like(explored_mask)
            mask = cv.drawContours(mask, [contour], 0, 1, -1)
            masked_values = unexplored_mask[mask.astype(bool)]
            values = set(masked_values.tolist())
            if 1 in values and len(values) == 1:
                small_contours.append(contour)
    new_explored_mask = explored_mask.copy()
    cv.drawContours(new_explored_mask, small_contours, -1, 255, -1)
    return new_explored_mask


def _bresenhamline(start: np.ndarray, end: np.ndarray) -> np.ndarray:
    """The N-D 'bresenhamline' recipe upstream uses with max_iter=-1: points start + k*slope/max|slope|, k=1..max|slope|,
    rounded half-even; the start point itself is not emitted."""
    max_iter = int(np.amax(np.abs(end - start)))
    if max_iter == 0:
        return np.zeros((0, start.shape[-1]), start.dtype)
    slope = (end - start).astype(np.double)
    nslope = slope / np.amax(np.abs(slope))
    steps = np.arange(1, max_iter + 1)[:, None]
    return np.array(np.rint(start[None, :] + nslope[None, :] * steps), dtype=start.dtype)


def interpolate_contour(contour: np.ndarray) -> np.ndarray:
    pts = contour.reshape(-1, 2)
    if len(pts) > 1:
        # CHAIN_APPROX_NONE contours (the only caller's): consecutive points are 8-neighbours, so every segment's
        # "bresenhamline" is exactly its end point -- the whole interpolation is the contour rotated by one
        step = np.abs(np.roll(pts, -1, axis=0) - pts).max(axis=1)
        if np.all(step == 1):
            return np.roll(pts, -1, axis=0).reshape((-1, 1, 2))
    segs = [_bresenhamline(pts[i], pts[(i + 1) % len(pts)]) for i in range(len(pts))]
    return np.concatenate(segs).reshape((-1, 1, 2)) if len(segs) else np.zeros((0, 1, 2), pts.dtype)


def contour_to_frontier_midpoints(contour: np.ndarray, unexplored_mask: np.ndarray) -> np.ndarray:
    """contour_to_frontiers + get_frontier_midpoint (numba-jitted upstream; fused here in oracle/cvport.c)."""
    c = np.ascontiguousarray(contour.reshape(-1, 2), np.int32)
    out = np.zeros((max(len(c), 1), 2), np.float64)
    n = cv.lib().cvp_contour_frontier_midpoints(c if len(c) else np.zeros((1, 2), np.int32), len(c),
                                                np.ascontiguousarray(unexplored_mask, np.uint8),
                                                unexplored_mask.shape[0], unexplored_mask.shape[1], out.reshape(-1),
                                                len(out))
    return out[:n]


def detect_frontier_waypoints(full_map: np.ndarray, explored_mask: np.ndarray, area_thresh: Optional[int] = -1,
                              xy: Optional[np.ndarray] = None) -> np.ndarray:
    assert xy is None
    explored_mask[full_map == 0] = 0
    filtered = filter_out_small_unexplored(full_map, explored_mask, area_thresh)
    contours, _ = cv.findContours(filtered, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_NONE)
    unexplored_mask = np.where(filtered > 0, 0, full_map)
    unexplored_mask = cv.blur(np.where(unexplored_mask > 0, 255, unexplored_mask).astype(np.uint8), (3, 3))
    waypoints: List[np.ndarray] = []
    for contour in contours:
        mids = contour_to_frontier_midpoints(interpolate_contour(contour), unexplored_mask)
        waypoints.extend(list(mids))
    return np.array(waypoints)

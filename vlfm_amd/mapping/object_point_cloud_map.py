"""vlfm.mapping.object_point_cloud_map.ObjectPointCloudMap, MI355X-native
(reference: /root/reference/vlfm/mapping/object_point_cloud_map.py:17-297).

The per-detection heavy lifting -- mask erosion, masked back-projection, DBSCAN + largest-cluster selection
(``_extract_object_cloud``, :150-170) -- runs in HIP kernels (csrc/object_cloud.hip); the bookkeeping around it (range ids,
closest point, ``get_best_object`` hysteresis, ``update_explored``) is O(points kept) NumPy on the host exactly as in the
reference, INCLUDING its draws from NumPy's global RNG (``np.random.choice`` for the 5000-point subsample,
``np.random.rand`` for the out-of-range ids), so that a seeded run reproduces the reference's clouds."""
from __future__ import annotations

from typing import Dict, Union

import numpy as np

from .. import _lib
from .base_map import require_gpu
from .value_map import _stream_ptr


def extract_yaw(matrix: np.ndarray) -> float:
    return float(np.arctan2(matrix[1, 0], matrix[0, 0]))  # geometry_utils.py:145-159


def transform_points(transformation_matrix: np.ndarray, points: np.ndarray) -> np.ndarray:
    """geometry_utils.py:205-213."""
    h = np.hstack((points, np.ones((points.shape[0], 1))))
    t = np.dot(transformation_matrix, h.T).T
    return t[:, :3] / t[:, 3:]


def within_fov_cone(cone_origin, cone_angle, cone_fov, cone_range, points) -> np.ndarray:
    """geometry_utils.py:91-116."""
    directions = points[:, :3] - cone_origin
    dists = np.linalg.norm(directions, axis=1)
    angles = np.arctan2(directions[:, 1], directions[:, 0])
    angle_diffs = np.mod(angles - cone_angle + np.pi, 2 * np.pi) - np.pi
    mask = np.logical_and(dists <= cone_range, np.abs(angle_diffs) <= cone_fov / 2)
    return points[mask]


def too_offset(mask: np.ndarray) -> bool:
    """object_point_cloud_map.py:272-297 (cv2.boundingRect of the non-zero pixels = their min/max column)."""
    cols = np.flatnonzero(np.asarray(mask).any(axis=0))
    if len(cols) == 0:
        x, w = 0, 0
    else:
        x, w = int(cols[0]), int(cols[-1] - cols[0] + 1)
    third = mask.shape[1] // 3
    if x + w <= third:
        return x <= int(0.05 * mask.shape[1])
    elif x >= 2 * third:
        return x + w >= int(0.95 * mask.shape[1])
    return False


def get_random_subarray(points, size: int):
    """object_point_cloud_map.py:253-269: the reference's own RNG call decides the indices."""
    if len(points) <= size:
        return points
    return points[np.random.choice(len(points), size, replace=False)]


class ObjectPointCloudMap:
    clouds: Dict[str, np.ndarray] = {}
    use_dbscan: bool = True

    def __init__(self, erosion_size: float, device=None) -> None:
        self._erosion_size = erosion_size
        self.last_target_coord: Union[np.ndarray, None] = None
        self.device = require_gpu(device)
        _lib.lib()
        self.clouds = {}
        self._bufs = None

    def reset(self) -> None:
        self.clouds = {}
        self.last_target_coord = None

    def has_object(self, target_class: str) -> bool:
        return target_class in self.clouds and len(self.clouds[target_class]) > 0

    # ------------------------------------------------------------------------------------------ :29-75
    def update_map(self, object_name: str, depth_img: np.ndarray, object_mask: np.ndarray,
                   tf_camera_to_episodic: np.ndarray, min_depth: float, max_depth: float, fx: float, fy: float) -> None:
        local_cloud = self._extract_object_cloud(depth_img, object_mask, min_depth, max_depth, fx, fy)
        if len(local_cloud) == 0:
            return
        if too_offset(object_mask):
            within_range = np.ones_like(local_cloud[:, 0]) * np.random.rand()
        else:
            within_range = (local_cloud[:, 0] <= max_depth * 0.95) * 1.0  # 5% margin
            within_range = within_range.astype(np.float32)
            within_range[within_range == 0] = np.random.rand()
        global_cloud = transform_points(tf_camera_to_episodic, local_cloud)
        global_cloud = np.concatenate((global_cloud, within_range[:, None]), axis=1)
        curr_position = tf_camera_to_episodic[:3, 3]
        closest_point = self._get_closest_point(global_cloud, curr_position)
        if np.linalg.norm(closest_point[:3] - curr_position) < 1.0:
            return  # too close to trust
        if object_name in self.clouds:
            self.clouds[object_name] = np.concatenate((self.clouds[object_name], global_cloud), axis=0)
        else:
            self.clouds[object_name] = global_cloud

    # ------------------------------------------------------------------------------------------ :77-101
    def get_best_object(self, target_class: str, curr_position: np.ndarray) -> np.ndarray:
        target_cloud = self.get_target_cloud(target_class)
        closest_point_2d = self._get_closest_point(target_cloud, curr_position)[:2]
        if self.last_target_coord is None:
            self.last_target_coord = closest_point_2d
        else:
            delta_dist = np.linalg.norm(closest_point_2d - self.last_target_coord)
            if delta_dist < 0.1:
                return self.last_target_coord
            elif delta_dist < 0.5 and np.linalg.norm(curr_position - closest_point_2d) > 2.0:
                return self.last_target_coord
            else:
                self.last_target_coord = closest_point_2d
        return self.last_target_coord

    # ------------------------------------------------------------------------------------------ :103-135
    def update_explored(self, tf_camera_to_episodic: np.ndarray, max_depth: float, cone_fov: float) -> None:
        camera_coordinates = tf_camera_to_episodic[:3, 3]
        camera_yaw = extract_yaw(tf_camera_to_episodic)
        for obj in self.clouds:
            within_range = within_fov_cone(camera_coordinates, camera_yaw, cone_fov, max_depth * 0.5, self.clouds[obj])
            for range_id in set(within_range[..., -1].tolist()):
                if range_id == 1:
                    continue
                self.clouds[obj] = self.clouds[obj][self.clouds[obj][..., -1] != range_id]

    def get_target_cloud(self, target_class: str) -> np.ndarray:
        target_cloud = self.clouds[target_class].copy()
        if np.any(target_cloud[:, -1] == 1):
            target_cloud = target_cloud[target_cloud[:, -1] == 1]
        return target_cloud

    # ------------------------------------------------------------------------------------------ :150-170 on the GPU
    def _extract_object_cloud(self, depth: np.ndarray, object_mask: np.ndarray, min_depth: float, max_depth: float,
                              fx: float, fy: float) -> np.ndarray:
        import torch

        L = _lib.lib()
        dev = self.device
        d = torch.from_numpy(np.ascontiguousarray(depth, np.float32)).to(dev) if not torch.is_tensor(depth) else depth
        d = d.reshape(d.shape[-2], d.shape[-1]).contiguous()
        H, W = d.shape
        m = torch.from_numpy(np.ascontiguousarray(np.asarray(object_mask) != 0).astype(np.uint8)).to(dev) \
            if not torch.is_tensor(object_mask) else (object_mask != 0).to(torch.uint8).contiguous()
        cap = H * W
        scratch = torch.empty(L.vlfm_object_cloud_scratch_bytes(H, W), dtype=torch.uint8, device=dev)
        cloud = torch.empty((cap, 3), dtype=torch.float64, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.vlfm_object_cloud_extract(d.data_ptr(), m.data_ptr(), H, W, int(self._erosion_size),
                                                   float(min_depth), float(max_depth), float(fx), float(fy),
                                                   scratch.data_ptr(), cloud.data_ptr(), cap, count.data_ptr(),
                                                   _stream_ptr()), "object_cloud_extract")
            n = int(count.item())
            cloud = cloud[:n]
            if n > 5000:  # get_random_subarray: NumPy's global RNG picks, the device gathers
                idx = np.random.choice(n, 5000, replace=False)
                cloud = cloud[torch.from_numpy(idx).to(dev)].contiguous()
                n = 5000
            if not self.use_dbscan or n == 0:
                return cloud.cpu().numpy()
            sc = torch.empty(L.vlfm_dbscan_scratch_bytes(n), dtype=torch.uint8, device=dev)
            labels = torch.empty(n, dtype=torch.int32, device=dev)
            keep = torch.empty(n, dtype=torch.int32, device=dev)
            num = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(L.vlfm_dbscan_largest_cluster(cloud.data_ptr(), n, 0.2, 100, sc.data_ptr(), sc.numel(),
                                                     labels.data_ptr(), keep.data_ptr(), num.data_ptr(), _stream_ptr()),
                       "dbscan")
            k = int(num.item())
            if k == 0:
                return np.array([])  # only noise was detected (:200-201)
            return cloud[keep[:k].to(torch.int64)].cpu().numpy()

    # ------------------------------------------------------------------------------------------ :172-183
    def _get_closest_point(self, cloud: np.ndarray, curr_position: np.ndarray) -> np.ndarray:
        ndim = curr_position.shape[0]
        if self.use_dbscan:
            return cloud[np.argmin(np.linalg.norm(cloud[:, :ndim] - curr_position, axis=1))]
        ref_point = np.concatenate((curr_position, np.array([0.5]))) if ndim == 2 else curr_position
        distances = np.linalg.norm(cloud[:, :3] - ref_point, axis=1)
        sorted_indices = np.argsort(distances)
        top_percent = sorted_indices[: int(0.25 * len(cloud))]
        try:
            median_index = top_percent[int(len(top_percent) / 2)]
        except IndexError:
            median_index = 0
        return cloud[median_index]

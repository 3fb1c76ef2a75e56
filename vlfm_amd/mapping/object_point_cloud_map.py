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


def too_offset(mask) -> bool:
    """Does the detection hug the left or right image border (object_point_cloud_map.py:272-297)?  A mask whose column
    extent lies wholly in the outer third of the image AND reaches within 5 % of that border is "too offset": its
    points get an out-of-range tag because the object is probably cut off by the image edge.  ``mask``: (H,W) ndarray, or a
    device tensor (the column occupancy is reduced on the device; W bytes cross to the host)."""
    width = mask.shape[1]
    if hasattr(mask, "is_cuda"):
        occupied = np.flatnonzero((mask != 0).any(dim=0).cpu().numpy())
    else:
        occupied = np.flatnonzero(np.asarray(mask).any(axis=0))       # cv2.boundingRect's x-extent
    left, right = (int(occupied[0]), int(occupied[-1]) + 1) if len(occupied) else (0, 0)
    band = width // 3
    if right <= band:
        return left <= int(0.05 * width)
    if left >= 2 * band:
        return right >= int(0.95 * width)
    return False


def get_random_subarray(points, size: int, rng=np.random):
    """At most ``size`` rows, chosen by NumPy's GLOBAL generator like the reference (:253-269), so a seeded session
    reproduces its clouds."""
    n = len(points)
    return points if n <= size else points[rng.choice(n, size, replace=False)]


class ObjectPointCloudMap:
    """Per object class: an (N, 4) cloud of episodic-frame points whose 4th column is a TAG -- 1.0 for points seen within
    95 % of the depth range, otherwise one random number per observation (so that all far points of an observation can be
    dropped together once the robot looks there from nearby, ``update_explored``)."""

    clouds: Dict[str, np.ndarray] = {}
    use_dbscan: bool = True

    def __init__(self, erosion_size: float, device=None, rng=None) -> None:
        """``rng``: where the two random draws come from.  Default = NumPy's GLOBAL generator, like the reference; the batched
        harness gives every environment its own ``np.random.RandomState(seed)`` (same stream as ``np.random.seed(seed)``
        followed by the global calls), so that E interleaved episodes each reproduce their single-environment run."""
        self._erosion_size = erosion_size
        self._rng = rng if rng is not None else np.random
        self.last_target_coord: Union[np.ndarray, None] = None
        self.device = require_gpu(device)
        _lib.lib()
        self.clouds = {}
        self._bufs = None
        self._near_only: Dict[str, tuple] = {}     # name -> (id(cloud array), rows): every tag of that array is 1.0

    def reset(self) -> None:
        self.clouds, self.last_target_coord = {}, None
        self._near_only = {}

    def _all_near(self, name: str) -> bool:
        """True when the cloud stored under ``name`` is known to hold only trusted (tag 1.0) points -- bookkeeping of update_map /
        update_explored, valid only for the very array it was recorded for (``clouds`` is a public dict)."""
        cloud = self.clouds.get(name)
        return cloud is not None and self._near_only.get(name) == (id(cloud), len(cloud))

    def has_object(self, target_class: str) -> bool:
        return len(self.clouds.get(target_class, ())) > 0

    def update_map(self, object_name: str, depth_img: np.ndarray, object_mask: np.ndarray,
                   tf_camera_to_episodic: np.ndarray, min_depth: float, max_depth: float, fx: float, fy: float) -> None:
        """object_point_cloud_map.py:29-75."""
        camera_frame = self._extract_object_cloud(depth_img, object_mask, min_depth, max_depth, fx, fy)
        if len(camera_frame) == 0:
            return
        # exactly ONE draw from the global generator per non-empty observation, whichever branch is taken (the reference
        # evaluates np.random.rand() on both paths); the "near" path keeps its tags in f32 like the reference's astype
        tag = self._rng.rand()
        if too_offset(object_mask):
            tags = np.full(len(camera_frame), tag, dtype=camera_frame.dtype)
        else:
            near = camera_frame[:, 0] <= max_depth * 0.95
            tags = np.where(near, np.float32(1.0), np.float32(tag)).astype(np.float32)
        tagged = np.concatenate((transform_points(tf_camera_to_episodic, camera_frame), tags[:, None]), axis=1)
        here = tf_camera_to_episodic[:3, 3]
        if np.linalg.norm(self._get_closest_point(tagged, here)[:3] - here) < 1.0:
            return  # closer than 1 m: depth this near is not trusted (:63-67)
        known = self.clouds.get(object_name)
        near_only = (known is None or self._all_near(object_name)) and bool(np.all(tags == 1))
        self.clouds[object_name] = tagged if known is None else np.concatenate((known, tagged), axis=0)
        if near_only:
            self._near_only[object_name] = (id(self.clouds[object_name]), len(self.clouds[object_name]))
        else:
            self._near_only.pop(object_name, None)

    def get_best_object(self, target_class: str, curr_position: np.ndarray) -> np.ndarray:
        """Goal point with hysteresis (:77-101): keep the previous goal when the new closest point moved < 0.1 m, or
        < 0.5 m while the robot is still more than 2 m away."""
        # (a cloud of trusted points only IS its target cloud: no copy, no mask -- the harness asks this for every environment
        # and step, on clouds of 10^5 points)
        cloud = self.clouds[target_class] if self._all_near(target_class) else self.get_target_cloud(target_class)
        candidate = self._get_closest_point(cloud, curr_position)[:2]
        previous = self.last_target_coord
        if previous is not None:
            moved = np.linalg.norm(candidate - previous)
            if moved < 0.1 or (moved < 0.5 and np.linalg.norm(curr_position - candidate) > 2.0):
                return previous
        self.last_target_coord = candidate
        return candidate

    def update_explored(self, tf_camera_to_episodic: np.ndarray, max_depth: float, cone_fov: float) -> None:
        """Forget far-tagged observations the robot is now looking at from within half the depth range (:103-135): every
        tag other than 1 that shows up inside the view cone is removed from the cloud as a whole."""
        origin, heading = tf_camera_to_episodic[:3, 3], extract_yaw(tf_camera_to_episodic)
        for name, cloud in list(self.clouds.items()):
            if self._all_near(name):
                continue       # only tags other than 1 can be dropped (`- {1}` below): nothing to do, same result
            seen_tags = set(within_fov_cone(origin, heading, cone_fov, max_depth * 0.5, cloud)[..., -1].tolist()) - {1}
            for t in seen_tags:
                cloud = cloud[cloud[..., -1] != t]
            self.clouds[name] = cloud

    def get_target_cloud(self, target_class: str) -> np.ndarray:
        """The trusted (tag 1) points when there are any, else everything (:137-144)."""
        cloud = self.clouds[target_class].copy()
        trusted = cloud[:, -1] == 1
        return cloud[trusted] if np.any(trusted) else cloud

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
                idx = self._rng.choice(n, 5000, replace=False)
                cloud = cloud[torch.from_numpy(idx).to(dev)].contiguous()
                n = 5000
            if not self.use_dbscan:
                return cloud.cpu().numpy()
            if n == 0:
                return np.array([])  # open3d_dbscan_filtering of an empty cloud: no non-noise label (:200-201), shape (0,) like the reference
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
        """With DBSCAN-cleaned clouds: the nearest point.  Without: the median of the nearest quarter, measured from the
        position lifted to 0.5 m when it is 2-D (outlier-tolerant)."""
        k = curr_position.shape[0]
        if self.use_dbscan:
            return cloud[np.argmin(np.linalg.norm(cloud[:, :k] - curr_position, axis=1))]
        anchor = curr_position if k != 2 else np.concatenate((curr_position, np.array([0.5])))
        order = np.argsort(np.linalg.norm(cloud[:, :3] - anchor, axis=1))
        nearest_quarter = order[: int(0.25 * len(cloud))]
        pick = nearest_quarter[int(len(nearest_quarter) / 2)] if len(nearest_quarter) else 0
        return cloud[pick]

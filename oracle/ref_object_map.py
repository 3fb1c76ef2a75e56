"""oracle/ref_object_map.py -- TEST INFRASTRUCTURE ONLY.

Restatements of the two third-party operations behind ObjectPointCloudMap._extract_object_cloud
(/root/reference/vlfm/mapping/object_point_cloud_map.py:150-170,186-212) that are not installable here:
cv2.erode(mask, None, iterations) [ext OpenCV] and open3d.geometry.PointCloud.cluster_dbscan [ext Open3D], the latter as the
SEQUENTIAL algorithm Open3D implements (scan in index order, breadth-first cluster growth, noise points may be re-labelled
as border points by the first cluster that reaches them).  PARITY UNPINNED against the real packages."""
from __future__ import annotations

import numpy as np


def erode3x3(mask_u8: np.ndarray, iterations: int) -> np.ndarray:
    """cv2.erode(mask, None, iterations=k): 3x3 rectangular minimum; pixels outside the image do not erode."""
    m = (np.asarray(mask_u8) != 0)
    for _ in range(int(iterations)):
        p = np.pad(m, 1, constant_values=True)
        out = np.ones_like(m)
        for dy in range(3):
            for dx in range(3):
                out &= p[dy:dy + m.shape[0], dx:dx + m.shape[1]]
        m = out
    return (m * 255).astype(np.uint8)


def cluster_dbscan(points: np.ndarray, eps: float, min_points: int) -> np.ndarray:
    """open3d cluster_dbscan: labels (-1 = noise, clusters 0.. in discovery order)."""
    n = len(points)
    p = np.asarray(points, np.float64)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2)
    d2 = d2[..., 0] + d2[..., 1] + d2[..., 2]
    nbs = [np.flatnonzero(d2[i] < eps * eps) for i in range(n)]  # radius search: strict, includes the point itself
    labels = np.full(n, -2, np.int64)
    cluster = 0
    for idx in range(n):
        if labels[idx] != -2:
            continue
        if len(nbs[idx]) < min_points:
            labels[idx] = -1
            continue
        todo = list(nbs[idx])
        queued = set(todo) | {idx}
        labels[idx] = cluster
        while todo:
            nb = todo.pop(0)
            if labels[nb] == -1:
                labels[nb] = cluster
            if labels[nb] != -2:
                continue
            labels[nb] = cluster
            if len(nbs[nb]) >= min_points:
                for q in nbs[nb]:
                    if q not in queued:
                        queued.add(q)
                        todo.append(q)
        cluster += 1
    return labels


def dbscan_filtering(points: np.ndarray, eps: float = 0.2, min_points: int = 100) -> np.ndarray:
    """object_point_cloud_map.py:186-212."""
    labels = cluster_dbscan(points, eps, min_points)
    unique_labels, label_counts = np.unique(labels, return_counts=True)
    keep = unique_labels != -1
    if not keep.any():
        return np.array([])
    largest = unique_labels[keep][np.argmax(label_counts[keep])]
    return points[np.where(labels == largest)[0]]


def extract_object_cloud(depth, object_mask, erosion, min_depth, max_depth, fx, fy, use_dbscan=True):
    """object_point_cloud_map.py:150-170 with the restated third-party calls."""
    from .ref_geometry import unproject

    final_mask = erode3x3(np.asarray(object_mask) * 255, erosion)
    valid = depth.copy()
    valid[valid == 0] = 1
    valid = valid * (max_depth - min_depth) + min_depth
    cloud = unproject(valid, final_mask, fx, fy)
    if len(cloud) > 5000:
        cloud = cloud[np.random.choice(len(cloud), 5000, replace=False)]
    return dbscan_filtering(cloud) if use_dbscan else cloud

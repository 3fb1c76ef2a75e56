"""Host-side mirror of vlfm.mapping.base_map.BaseMap (/root/reference/vlfm/mapping/base_map.py:10-60).

Holds only the coordinate conventions and trajectory bookkeeping; the map payload itself lives in HBM
(see value_map.py / obstacle_map.py).  Pure NumPy scalar work -- nothing here is on the per-pixel path.
"""
from __future__ import annotations

from typing import Any, List

import numpy as np


def require_gpu(device=None):
    """vlfm_amd has no CPU execution path: fail loudly when there is no HIP device."""
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError(
            "vlfm_amd needs an AMD Instinct GPU (gfx950) visible to PyTorch-ROCm; there is no CPU fallback. "
            "(The NumPy oracle under oracle/ is test infrastructure, not a backend.)")
    return torch.device(device if device is not None else "cuda:0")


class BaseMap:
    """Coordinate conventions and trajectory bookkeeping of the reference's BaseMap (same constructor, same attribute
    names); the ``_map`` payload is provided by the subclasses from HBM."""

    _camera_positions: List[np.ndarray] = []
    _last_camera_yaw: float = 0.0

    def __init__(self, size: int = 1000, pixels_per_meter: int = 20, *args: Any, **kwargs: Any):
        self.size, self.pixels_per_meter = size, pixels_per_meter
        self._episode_pixel_origin = np.array([size // 2, size // 2])   # the episode starts at the map centre
        self._camera_positions = []

    def reset(self) -> None:
        self._camera_positions = []

    def update_agent_traj(self, robot_xy: np.ndarray, robot_heading: float) -> None:
        self._last_camera_yaw = robot_heading
        self._camera_positions.append(robot_xy)

    def _xy_to_px(self, points: np.ndarray) -> np.ndarray:
        """Metres (x forward, y left) -> integer (col, row) cells: the axes swap, ``rint`` rounds half to even, and the
        first coordinate is mirrored about the map size (base_map.py:44-46)."""
        cells = np.rint(points[:, ::-1] * self.pixels_per_meter) + self._episode_pixel_origin
        cells[:, 0] = self.size - cells[:, 0]
        return cells.astype(int)

    def _px_to_xy(self, px: np.ndarray) -> np.ndarray:
        """Inverse of ``_xy_to_px`` without the rounding (base_map.py:57-60)."""
        unmirrored = np.array(px, copy=True)
        unmirrored[:, 0] = self.size - unmirrored[:, 0]
        return ((unmirrored - self._episode_pixel_origin) / self.pixels_per_meter)[:, ::-1]

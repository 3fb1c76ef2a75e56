"""Synthetic RGB-D episodes of SURVEY.md section 8(d): deterministic (PCG64, seed = 1234 + env_id) depth frames,
scripted trajectories and stub values.  Used by bench.py, the harness and the parity tests (both the HIP path and the
oracle are fed from here, so they see identical inputs)."""
from __future__ import annotations

import numpy as np

HFOV_DEG = 79.0
MIN_DEPTH, MAX_DEPTH = 0.5, 5.0
CAMERA_HEIGHT = 0.88


def camera_intrinsics(width: int, hfov_deg: float = HFOV_DEG):
    """fx = fy = W / (2 tan(hfov/2))  (/root/reference/vlfm/policy/habitat_policies.py:91)."""
    hfov = np.deg2rad(hfov_deg)
    fx = width / (2 * np.tan(hfov / 2))
    return float(fx), float(fx), float(hfov)


def depth_frame(rng: np.random.Generator, height: int = 480, width: int = 640, holes: bool = False) -> np.ndarray:
    """Piecewise-smooth wall profile + floor plane, normalised to (0,1] (no exact zeros unless holes=True)."""
    fx, fy, _ = camera_intrinsics(width)
    wall = rng.uniform(1.0, 5.0, size=width + 30)
    wall = np.convolve(wall, np.ones(31) / 31.0, mode="valid")[:width]
    rows = np.arange(height)[:, None] - height // 2
    with np.errstate(divide="ignore"):
        floor = np.where(rows > 0, CAMERA_HEIGHT * fy / np.maximum(rows, 1e-9), np.inf)
    d = np.minimum(wall[None, :], floor)
    d = np.clip((d - MIN_DEPTH) / (MAX_DEPTH - MIN_DEPTH), 1e-3, 1.0).astype(np.float32)
    if holes:
        for _ in range(5):
            r0, c0 = rng.integers(0, height - 40), rng.integers(0, width - 40)
            d[r0:r0 + rng.integers(5, 40), c0:c0 + rng.integers(5, 40)] = 0.0
    return d


def rgb_frame(rng: np.random.Generator, height: int = 480, width: int = 640) -> np.ndarray:
    return rng.integers(0, 256, size=(height, width, 3), dtype=np.uint8)


def pose_to_tf(x: float, y: float, yaw: float, z: float = CAMERA_HEIGHT) -> np.ndarray:
    """geometry_utils.py:162-180 (xyz_yaw_to_tf_matrix)."""
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0, x], [s, c, 0, y], [0, 0, 1, z], [0, 0, 0, 1]])


class Trajectory:
    """Steps 0-11 turn 30 deg (habitat_policies.py:150-153), then forward 0.25 m w.p. 0.7 / turn +-30 deg."""

    def __init__(self, env_id: int, limit: float = 20.0) -> None:
        self.rng = np.random.Generator(np.random.PCG64(1234 + env_id))
        self.x = self.y = self.yaw = 0.0
        self.t = 0
        self.limit = limit

    def step(self):
        if self.t > 0:
            if self.t <= 11:
                self.yaw += np.deg2rad(30)
            else:
                u = self.rng.uniform()
                if u < 0.7:
                    nx, ny = self.x + 0.25 * np.cos(self.yaw), self.y + 0.25 * np.sin(self.yaw)
                    if abs(nx) <= self.limit and abs(ny) <= self.limit:
                        self.x, self.y = nx, ny
                    else:
                        self.yaw += np.deg2rad(30)
                elif u < 0.85:
                    self.yaw += np.deg2rad(30)
                else:
                    self.yaw -= np.deg2rad(30)
        self.t += 1
        yaw = (self.yaw + np.pi) % (2 * np.pi) - np.pi
        return self.x, self.y, yaw


class SyntheticEnv:
    def __init__(self, env_id: int, height: int = 480, width: int = 640, holes: bool = False, channels: int = 1):
        self.traj = Trajectory(env_id)
        self.rng = np.random.Generator(np.random.PCG64(99991 + env_id))
        self.height, self.width, self.holes, self.channels = height, width, holes, channels

    def observe(self):
        x, y, yaw = self.traj.step()
        depth = depth_frame(self.rng, self.height, self.width, self.holes)
        values = self.rng.uniform(0.15, 0.45, size=self.channels)
        return depth, pose_to_tf(x, y, yaw), values

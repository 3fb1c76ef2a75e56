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


# ---------------------------------------------------------------------------------------------- a consistent world
# Rooms, doorways, free-standing pillars, an L-shaped block and a U-shaped alcove, rendered by per-column ray casting.
# Everything that produces an observation uses only IEEE-exact operations (+ - * / sqrt, comparisons): headings are
# multiples of 30 degrees whose cosines / sines come from a sqrt(3)/2 table, a ray through image column u has the
# camera-frame direction (1, -(u - W/2)/fx) -- so the ray parameter IS the depth along the optical axis -- and poses are
# integrated from an action string (turn 30 degrees / forward 0.25 m: the reference's simulator step sizes,
# vlfm/policy/action_replay_policy.py:41-42).  Used by the 500-step golden episode (tests/golden/world500.py), the
# config-5 parity test and the batched-episode harness (frontier-rich maps instead of the per-frame random walls above).
ROOMS_H, ROOMS_W = 480, 640
ROOMS_STEPS = 500
_S3 = float(np.sqrt(3.0) / 2.0)
# (cos, sin) of k * 30 degrees, k = 0..11 -- sqrt is correctly rounded everywhere
HEADINGS = [(1.0, 0.0), (_S3, 0.5), (0.5, _S3), (0.0, 1.0), (-0.5, _S3), (-_S3, 0.5),
            (-1.0, 0.0), (-_S3, -0.5), (-0.5, -_S3), (0.0, -1.0), (0.5, -_S3), (_S3, -0.5)]
YAWS = [k * np.pi / 6 if k <= 6 else (k - 12) * np.pi / 6 for k in range(12)]  # reported to the policy layer only
LEFT, RIGHT, FORWARD = 0, 1, 2


def _walls():
    t = 0.3
    b = [(-10, -10, 10, -10 + t), (-10, 10 - t, 10, 10), (-10, -10, -10 + t, 10), (10 - t, -10, 10, 10)]  # boundary
    # central hall 8.6 m x 8.6 m, two 1.2 m doorways per side
    for y0, y1 in ((4.0, 4.3), (-4.3, -4.0)):
        for x0, x1 in ((-4.3, -2.6), (-1.4, 1.4), (2.6, 4.3)):
            b.append((x0, y0, x1, y1))                      # north / south wall segments
            b.append((y0, x0, y1, x1))                      # west / east wall segments (transposed)
    # partition walls of the ring around the hall, each leaving a gap
    b += [(4.3, 5.5, 7.8, 5.8), (-0.15, 4.3, 0.15, 7.5), (-9.7, 5.0, -6.2, 5.3), (-7.5, -4.3, -7.2, -0.5),
          (-5.5, -9.7, -5.2, -6.0), (1.0, -7.0, 4.5, -6.7), (6.5, -9.7, 6.8, -6.5)]
    # free-standing pillars (the explored area closes around them)
    for cx, cy, r in ((1.6, 1.2, 0.3), (-1.8, 2.0, 0.35), (-2.2, -1.5, 0.3), (1.9, -2.1, 0.4), (7.6, 2.6, 0.45),
                      (-6.0, 8.0, 0.5), (4.0, 8.2, 0.4), (-8.5, -7.5, 0.5), (8.3, -4.5, 0.4)):
        b.append((cx - r, cy - r, cx + r, cy + r))
    b += [(6.0, -1.5, 8.5, -1.0), (6.0, -1.5, 6.5, 0.8)]                                   # L-shaped block (non-convex)
    b += [(-3.6, 6.3, -3.3, 8.6), (-3.6, 8.3, -1.6, 8.6), (-1.9, 6.3, -1.6, 8.6)]          # U-shaped alcove
    return np.array(b, np.float64)


BOXES = _walls()
# tour: hall -> east door -> north-east room -> north room (around the alcove) -> west rooms -> back through the west door ->
# south door -> south-east rooms
WAYPOINTS = [(0.6, 2.0), (3.4, 2.0), (5.4, 2.0), (6.9, 3.9), (8.9, 4.2), (8.9, 7.0), (6.0, 7.3), (2.2, 7.0), (1.0, 8.6), (-0.9, 8.7),
             (-0.9, 5.4), (-2.7, 5.3), (-4.9, 5.6), (-5.3, 4.6), (-8.6, 3.6), (-8.6, 0.5), (-5.6, 0.6), (-5.6, -2.0),
             (-3.4, -2.0), (-3.2, -3.0), (0.0, -3.0), (2.0, -3.3), (2.0, -5.6), (5.4, -5.8), (8.6, -5.9), (8.6, -8.5),
             (8.6, -5.9), (9.2, -5.7), (9.2, -2.2), (5.4, -2.0), (3.4, -2.0), (3.0, -0.6), (0.0, 0.0)]


def wall_profile(x: float, y: float, k: int, width: int = ROOMS_W) -> np.ndarray:
    """(width,) f32 depth along the optical axis of the nearest box face per image column (inf where nothing is hit)."""
    fx = camera_intrinsics(width)[0]
    c, s = HEADINGS[k]
    m = -(np.arange(width, dtype=np.float64) - width // 2) / fx        # geometry_utils.py:216-236: y_cam = -(u - W//2) z / fx
    dx, dy = (c - s * m)[:, None], (s + c * m)[:, None]        # world direction of (1, m): parameter t == z
    tiny = 1e-12
    dx = np.where(np.abs(dx) < tiny, tiny, dx)
    dy = np.where(np.abs(dy) < tiny, tiny, dy)
    tx0, tx1 = (BOXES[None, :, 0] - x) / dx, (BOXES[None, :, 2] - x) / dx
    ty0, ty1 = (BOXES[None, :, 1] - y) / dy, (BOXES[None, :, 3] - y) / dy
    tmin = np.maximum(np.minimum(tx0, tx1), np.minimum(ty0, ty1))
    tmax = np.minimum(np.maximum(tx0, tx1), np.maximum(ty0, ty1))
    hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)
    return np.where(hit, tmin, np.inf).min(axis=1).astype(np.float32)


def depth_from_profile(wall: np.ndarray, height: int = ROOMS_H) -> np.ndarray:
    fy = camera_intrinsics(len(wall))[1]
    rows = np.arange(height)[:, None] - height // 2
    floor = np.where(rows > 0, CAMERA_HEIGHT * fy / np.maximum(rows, 1e-9), np.inf)
    d = np.minimum(wall.astype(np.float64)[None, :], floor)
    return np.clip((d - MIN_DEPTH) / (MAX_DEPTH - MIN_DEPTH), 1e-3, 1.0).astype(np.float32)


def tf_of(x: float, y: float, k: int) -> np.ndarray:
    """xyz_yaw_to_tf_matrix (geometry_utils.py:162-180) with the exact (cos, sin) of the heading table."""
    c, s = HEADINGS[k]
    return np.array([[c, -s, 0.0, x], [s, c, 0.0, y], [0.0, 0.0, 1.0, CAMERA_HEIGHT], [0.0, 0.0, 0.0, 1.0]])


def _blocked(x: float, y: float, margin: float = 0.3) -> bool:
    return bool(np.any((BOXES[:, 0] - margin <= x) & (x <= BOXES[:, 2] + margin) &
                       (BOXES[:, 1] - margin <= y) & (y <= BOXES[:, 3] + margin)))


def integrate(actions):
    """Poses (x, y, k) BEFORE each action: the observation of step i is taken at poses[i]."""
    x = y = 0.0
    k = 0
    poses = []
    for a in actions:
        poses.append((x, y, k))
        if a == LEFT:
            k = (k + 1) % 12
        elif a == RIGHT:
            k = (k - 1) % 12
        else:
            c, s = HEADINGS[k]
            x, y = x + 0.25 * c, y + 0.25 * s
    return poses


def plan_actions(steps: int = ROOMS_STEPS) -> np.ndarray:
    """12 initial left turns (habitat_policies.py:150-153), then a waypoint follower: face the heading of the 30-degree
    set that is best aligned with the next waypoint, step forward.  Deterministic, but only run by the generator --
    replayers integrate the stored action string."""
    for p, q in zip([(0.0, 0.0)] + WAYPOINTS[:-1], WAYPOINTS):  # the hand-placed legs must be collision-free
        for u in np.linspace(0.0, 1.0, int(np.hypot(q[0] - p[0], q[1] - p[1]) / 0.05) + 2):
            assert not _blocked(p[0] + u * (q[0] - p[0]), p[1] + u * (q[1] - p[1]), 0.4), (p, q)
    x = y = 0.0
    k = 0
    acts = []
    wp = 0
    while len(acts) < steps:
        if len(acts) < 12:
            a = LEFT
        else:
            tx, ty = WAYPOINTS[wp % len(WAYPOINTS)]
            if (tx - x) ** 2 + (ty - y) ** 2 < 0.3 ** 2:
                wp += 1
                continue
            best = max(range(12), key=lambda j: (HEADINGS[j][0] * (tx - x) + HEADINGS[j][1] * (ty - y), -j))
            turn = (best - k) % 12
            a = FORWARD if turn == 0 else (LEFT if turn <= 6 else RIGHT)
        acts.append(a)
        if a == LEFT:
            k = (k + 1) % 12
        elif a == RIGHT:
            k = (k - 1) % 12
        else:
            c, s = HEADINGS[k]
            x, y = x + 0.25 * c, y + 0.25 * s
            assert not _blocked(x, y, 0.2), (len(acts), x, y)
    return np.array(acts, np.uint8)



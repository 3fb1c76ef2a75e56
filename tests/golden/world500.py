"""A 500-step episode (the reference's episode length, config/tasks/pointnav_depth_hm3d.yaml:14) in a consistent 2-D
world of rooms, doorways, free-standing pillars and an L-shaped block, rendered by per-column ray casting.

Shared by tests/golden/make_golden.py (which drives THE REFERENCE'S OWN ObstacleMap + ValueMap with it) and by the
replaying tests, so both sides see byte-identical observations.  Everything that produces an observation uses only
IEEE-exact operations (+ - * / sqrt, comparisons): headings are multiples of 30 degrees whose cosines / sines come from
a sqrt(3)/2 table, a ray through image column u has the camera-frame direction (1, -(u - W/2)/fx) -- so the ray
parameter IS the depth along the optical axis -- and the pose is integrated from a stored action string.  A replayer
therefore regenerates the frames bit for bit on any machine (the fixture still carries a digest of every frame).

The tour is planned ONCE by plan_actions() (a waypoint follower over the action set of the reference's simulator
config: turn 30 degrees / forward 0.25 m, vlfm/policy/action_replay_policy.py:41-42) and stored in the fixture.
"""
from __future__ import annotations

import numpy as np

from vlfm_amd.synthetic import CAMERA_HEIGHT, MAX_DEPTH, MIN_DEPTH, camera_intrinsics

H, W = 480, 640
STEPS = 500
SEED = 31
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


def wall_profile(x: float, y: float, k: int, width: int = W) -> np.ndarray:
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


def depth_from_profile(wall: np.ndarray, height: int = H) -> np.ndarray:
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


def plan_actions(steps: int = STEPS) -> np.ndarray:
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


def episode(actions):
    """Yield (step, depth f32 (H,W), tf (4,4) f64, values (1,) f64) for every step of the stored action string."""
    rng = np.random.Generator(np.random.PCG64(SEED))
    for i, (x, y, k) in enumerate(integrate(actions)):
        yield i, depth_from_profile(wall_profile(x, y, k)), tf_of(x, y, k), rng.uniform(0.15, 0.45, size=1)

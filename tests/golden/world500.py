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

from vlfm_amd.synthetic import (BOXES, FORWARD, HEADINGS, LEFT, RIGHT, WAYPOINTS, YAWS, depth_from_profile, integrate,  # noqa: F401
                                plan_actions, tf_of, wall_profile)

H, W = 480, 640
STEPS = 500
SEED = 31


def episode(actions):
    """Yield (step, depth f32 (H,W), tf (4,4) f64, values (1,) f64) for every step of the stored action string."""
    rng = np.random.Generator(np.random.PCG64(SEED))
    for i, (x, y, k) in enumerate(integrate(actions)):
        yield i, depth_from_profile(wall_profile(x, y, k)), tf_of(x, y, k), rng.uniform(0.15, 0.45, size=1)

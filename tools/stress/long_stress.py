"""LONG random episodes in the CONSISTENT rooms-and-pillars world of vlfm_amd/synthetic.py (the large, ragged explored outlines late in an episode:
tens of thousands of border-walk states, the follower's tables beyond LDS): a random walk with ARBITRARY headings (not the 30-degree set of the 500-step
fixture), collision-checked, `steps` steps per seed; GPU map against the oracle after every step (obstacle plane, explored area, frontier pixels).
    python tools/stress/long_stress.py first_seed n_seeds [steps]"""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util, numpy as np, torch
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, "tests", "test_obstacle_map_gpu.py")); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
from vlfm_amd import synthetic as syn
from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, depth_from_profile, pose_to_tf


def wall_profile_any(x, y, yaw, width=640):
    """synthetic.wall_profile for an arbitrary heading (same ray casting; cos / sin from libm instead of the exact table)."""
    fx = syn.camera_intrinsics(width)[0]
    c, s = float(np.cos(yaw)), float(np.sin(yaw))
    m = -(np.arange(width, dtype=np.float64) - width // 2) / fx
    dx, dy = (c - s * m)[:, None], (s + c * m)[:, None]
    dx = np.where(np.abs(dx) < 1e-12, 1e-12, dx); dy = np.where(np.abs(dy) < 1e-12, 1e-12, dy)
    B = syn.BOXES
    tx0, tx1 = (B[None, :, 0] - x) / dx, (B[None, :, 2] - x) / dx
    ty0, ty1 = (B[None, :, 1] - y) / dy, (B[None, :, 3] - y) / dy
    tmin = np.maximum(np.minimum(tx0, tx1), np.minimum(ty0, ty1)); tmax = np.minimum(np.maximum(tx0, tx1), np.maximum(ty0, ty1))
    hit = (tmax >= np.maximum(tmin, 0.0)) & (tmin > 0.0)
    return np.where(hit, tmin, np.inf).min(axis=1).astype(np.float32)
a, n = int(sys.argv[1]), int(sys.argv[2])
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 300
VALUES = "--values" in sys.argv        # also a ValueMap beside the obstacle map, and sort_waypoints over the step's frontiers
from oracle.ref_value_map import RefValueMap
from vlfm_amd.mapping import ValueMap
bad = 0
for seed in range(a, a + n):
    ours, ref = t._pair(torch.device("cuda:0"))
    vm = ValueMap(1, use_max_confidence=False, device=torch.device("cuda:0")) if VALUES else None
    rvm = RefValueMap(1, use_max_confidence=False) if VALUES else None
    rng = np.random.default_rng(90_000 + seed)
    x = y = 0.0
    nudged = skipped = 0
    try:
        for step in range(STEPS):
            yaw = rng.uniform(-np.pi, np.pi) if step % 5 else float(int(rng.integers(-6, 7)) * np.pi / 6)
            for _ in range(30):                                   # a collision-free move of up to 0.6 m
                nx, ny = x + rng.uniform(-0.6, 0.6), y + rng.uniform(-0.6, 0.6)
                if abs(nx) < 9.3 and abs(ny) < 9.3 and not syn._blocked(nx, ny, 0.35):
                    x, y = float(nx), float(ny); break
            d = depth_from_profile(wall_profile_any(x, y, yaw))
            tf = pose_to_tf(x, y, yaw)
            for m in (ours, ref): m.update_map(d, tf, MIN_DEPTH, MAX_DEPTH, t.FX, t.FY, t.FOV, explore=False)
            tf2 = tf
            for _ in range(20):
                if not t._extreme_angle_tie(ref, tf2): break
                nudged += 1
                tf2 = pose_to_tf(x + rng.uniform(-0.3, 0.3), y + rng.uniform(-0.3, 0.3), yaw)
            else:
                skipped += 1; continue
            for m in (ours, ref): m.update_map(None, tf2, MIN_DEPTH, MAX_DEPTH, t.FX, t.FY, t.FOV, update_obstacles=False)
            t._same(ours, ref, step)
            if VALUES:
                vals = rng.uniform(0.1, 0.5, 1)
                vm.update_map(vals, d, tf, MIN_DEPTH, MAX_DEPTH, t.FOV); rvm.update_map(vals, d.copy(), tf, MIN_DEPTH, MAX_DEPTH, t.FOV)
                fr = np.asarray(ref.frontiers, np.float64).reshape(-1, 2)
                if len(fr):
                    sa, sb = vm.sort_waypoints(fr, 0.5), rvm.sort_waypoints(fr, 0.5)
                    assert np.array_equal(sa[0], sb[0]) and np.array_equal(np.array(sa[1], float), np.array(sb[1], float)), f"sort_waypoints differs at step {step}"
                if step % 25 == 24 or step == STEPS - 1:
                    assert np.array_equal(vm._map, rvm._map) and np.array_equal(np.asarray(vm._value_map), np.asarray(rvm._value_map)), f"value map differs at step {step}"
        print(f"seed {seed}: {STEPS} steps equal; explored {int(ref.explored_area.sum())} cells, obstacles {int(ref._map.sum())}, frontiers {len(np.asarray(ref._frontiers_px).reshape(-1, 2))}, ties nudged {nudged}", flush=True)
    except AssertionError as e:
        bad += 1; print("seed", seed, "FAILED:", str(e)[:300].replace("\n", " "), flush=True)
    except Exception as e:
        bad += 1; print("seed", seed, "RAISED", type(e).__name__, str(e)[:300], flush=True)
print(f"long episodes, seeds {a}..{a + n - 1} x {STEPS} steps: {bad} failed")

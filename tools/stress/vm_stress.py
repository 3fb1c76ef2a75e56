import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle.ref_value_map import RefValueMap
from vlfm_amd.mapping import ValueMapBatch
from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, camera_intrinsics, pose_to_tf
fov = camera_intrinsics(640)[2]
seed0, rounds = int(sys.argv[1]), int(sys.argv[2])
E, H, W = 32, 24, 640
bad = 0
for rnd in range(rounds):
    rng = np.random.default_rng(seed0 + rnd)
    mode = rnd % 3
    batch = ValueMapBatch(E, 1, use_max_confidence=(mode == 1), fusion_type=("default", "default", "equal_weighting")[mode], device=torch.device("cuda:0"))
    refs = [RefValueMap(1, use_max_confidence=(mode == 1), fusion_type=("default", "default", "equal_weighting")[mode]) for _ in range(E)]
    poses = rng.uniform(-10, 10, (E, 2))
    for obs in range(3):
        depth = np.empty((E, H, W), np.float32); tf = np.empty((E, 4, 4))
        vals = rng.uniform(0.0, 0.7, (E, 1))
        for e in range(E):
            k = int(rng.integers(0, 5))
            prof = (rng.uniform(0, 1, W) if k == 0 else np.repeat(rng.uniform(0, 1, W // 8), 8) if k == 1 else
                    np.clip(np.cumsum(rng.normal(0, 0.05, W)) + rng.uniform(0.2, 0.8), 0, 1) if k == 2 else
                    np.where(rng.uniform(size=W) < rng.uniform(0.05, 0.95), rng.uniform(0, 0.3), rng.uniform(0.6, 1.0)) if k == 3 else
                    np.full(W, rng.uniform(0, 1)))
            d = rng.uniform(0, 1, (H, W)).astype(np.float32) * prof[None].astype(np.float32); d[0] = prof.astype(np.float32)
            depth[e] = d
            yaw = rng.uniform(-np.pi, np.pi) if rng.uniform() < 0.8 else float(np.nextafter(int(rng.integers(-8, 9)) * np.pi / 8, rng.choice([-10.0, 10.0])))
            p = poses[e] + rng.uniform(-0.4, 0.4, 2) * obs
            if rng.uniform() < 0.3: p = np.round(p * 20) / 20 + rng.choice([0.0, 1e-13, -1e-13])
            tf[e] = pose_to_tf(p[0], p[1], yaw)
        batch.update(vals, depth, tf, MIN_DEPTH, MAX_DEPTH, fov)
        for e in range(E): refs[e].update_map(vals[e], depth[e].copy(), tf[e], MIN_DEPTH, MAX_DEPTH, fov)
    conf = batch.conf.cpu().numpy(); val = batch.value.cpu().numpy()
    for e in range(E):
        if not (np.array_equal(conf[e], refs[e]._map) and np.array_equal(val[e].reshape(refs[e]._value_map.shape), refs[e]._value_map)):
            bad += 1
            print("round", rnd, "env", e, "mode", mode, "conf diff", int((conf[e] != refs[e]._map).sum()), "max", float(np.abs(conf[e] - refs[e]._map).max()))
print(f"value-map stress seeds {seed0}..{seed0 + rounds - 1} x {E} envs x 3 observations: {bad} maps differ")

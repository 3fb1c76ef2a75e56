"""ObstacleMapBatch + ValueMapBatch with E environments on INDEPENDENT random walks (what bench.py runs), resets of random slots at random steps:
after every step every slot's obstacle plane, explored area, frontier pixels and value map against that slot's own oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from oracle.ref_obstacle_map import RefObstacleMap
from oracle.ref_value_map import RefValueMap
from vlfm_amd.mapping import ObstacleMapBatch, ValueMapBatch
from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, camera_intrinsics, depth_frame, pose_to_tf
FX, FY, FOV = camera_intrinsics(640)
dev = torch.device("cuda:0")
a, b = int(sys.argv[1]), int(sys.argv[2])
E = 12
bad = 0
for seed in range(a, b):
    rng = np.random.default_rng(60_000 + seed)
    kw = dict(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5)
    ob = ObstacleMapBatch(E, device=dev, **kw)
    vb = ValueMapBatch(E, 1, use_max_confidence=False, device=dev)
    refs = [RefObstacleMap(**kw) for _ in range(E)]
    vrefs = [RefValueMap(1, use_max_confidence=False) for _ in range(E)]
    pos = rng.uniform(-8, 8, (E, 2))
    dropped = np.zeros(E, bool)
    ok = True
    for step in range(24):
        if step and step % 7 == 0:                      # reset a few slots mid-run
            idx = sorted(set(int(i) for i in rng.integers(0, E, 3)))
            ob.reset(idx); vb.reset(idx)
            for i in idx:
                refs[i].reset(); vrefs[i].reset(); dropped[i] = False
        depth = np.empty((E, 480, 640), np.float32); tf = np.empty((E, 4, 4))
        for e in range(E):
            pos[e] += rng.uniform(-0.5, 0.5, 2)
            d = depth_frame(rng)
            if step % 3 != 2: d[:] = np.maximum(d, np.float32(0.85))
            for _ in range(int(rng.integers(0, 5))):
                c0 = int(rng.integers(0, 600)); w = int(rng.integers(4, 120)); r0 = int(rng.integers(0, 300)); h = int(rng.integers(40, 480 - r0))
                d[r0:r0 + h, c0:c0 + w] = np.float32(rng.uniform(0.05, 0.7))
            depth[e] = d; tf[e] = pose_to_tf(pos[e, 0], pos[e, 1], rng.uniform(-np.pi, np.pi))
        vals = rng.uniform(0.05, 0.6, (E, 1))
        dd = torch.from_numpy(depth).to(dev)
        colmax = ob.ingest(dd, tf, MIN_DEPTH, MAX_DEPTH, FX, FY, want_colmax=True)
        ob.check_status()
        ob.update_after_ingest(tf, MAX_DEPTH, FOV)
        vb.update(vals, None, tf, MIN_DEPTH, MAX_DEPTH, FOV, colmax=colmax)
        fr = ob.frontiers_px()
        obst = ob._unpack(ob.obstacle_bits).cpu().numpy().astype(bool)
        expl = ob.explored.cpu().numpy().astype(bool)
        conf = vb.conf.cpu().numpy(); val = vb.value.cpu().numpy()
        for e in range(E):
            refs[e].update_map(depth[e].copy(), tf[e], MIN_DEPTH, MAX_DEPTH, FX, FY, FOV)
            vrefs[e].update_map(vals[e], depth[e].copy(), tf[e], MIN_DEPTH, MAX_DEPTH, FOV)
            if dropped[e]: continue
            if not np.array_equal(obst[e], refs[e]._map.astype(bool)):
                print("seed", seed, "step", step, "slot", e, "OBSTACLE plane differs"); ok = False
            elif not np.array_equal(expl[e], refs[e].explored_area.astype(bool)):
                print("seed", seed, "step", step, "slot", e, "explored differs by", int((expl[e] != refs[e].explored_area.astype(bool)).sum()), "cells (tie of the reference? slot dropped until its reset)")
                dropped[e] = True
            else:
                rf = np.asarray(refs[e]._frontiers_px, np.float64).reshape(-1, 2)
                if not np.array_equal(np.asarray(fr[e], np.float64).reshape(-1, 2), rf):
                    print("seed", seed, "step", step, "slot", e, "FRONTIERS differ"); ok = False
            if not (np.array_equal(conf[e], vrefs[e]._map) and np.array_equal(val[e].reshape(vrefs[e]._value_map.shape), vrefs[e]._value_map)):
                print("seed", seed, "step", step, "slot", e, "VALUE map differs"); ok = False
        if not ok: break
    bad += not ok
print(f"batched maps ({E} slots, resets), seeds {a}..{b - 1}: {bad} failed")

"""Where /root/reference exists: RANDOM scripted ObjectNav episodes through THE REFERENCE'S ITMPolicyV2 (vlfm/policy/itm_policy.py + base_objectnav_policy.py,
real source through oracle/ref_shim.py; the generator of tests/golden/make_golden.py run live, nothing written) and, step by step, through
vlfm_amd/policy_step.py:ITMPolicyV2Step over the reference's own map classes -- which isolates the decision path.  The five recorded episodes
(tests/golden/policy_*.npz) are hand-scripted; here the sightings are drawn at random: either detector, target and distractor phrases, confidences
around both thresholds, objects near, far, large, tiny or cut by the image border, on all three category kinds (COCO class / MP3D multi-name /
non-COCO).  Per step: mode, frontier list, pursued goal, (rho, theta), stop flag, best value, SAM mask pixels, controller resets; at the end the order
in which the models were asked, the prompts, and the maps."""
import os
import sys

import numpy as np
import pytest

import golden_util
from golden_util import GOLDEN_DIR, dense, replay_policy_episode, sha, unpack_plane

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/vlfm"), reason="needs the reference checkout")


def _random_script(ps, seed: int, steps: int):
    rng = np.random.default_rng(8000 + seed)
    kind = seed % 3
    if kind == 0:
        dataset, goal, caption = "hm3d", str(rng.choice(["chair", "bed", "tv", "potted plant", "toilet", "couch"])), ""
        targets, distract = [goal], ["dining table", "person", "clock"]
    elif kind == 1:
        dataset, goal, caption = "mp3d", "table|dining table|coffee table|side table|desk", ps.MP3D_CAPTION
        targets, distract = goal.split("|"), ["chair", "cabinet", "bed"]
    else:
        dataset, goal, caption = "mp3d", "cabinet", ps.MP3D_CAPTION
        targets, distract = ["cabinet"], ["chair", "table", "shelving"]
    sightings = {}
    for k in range(12, steps):
        if rng.uniform() > 0.4:
            continue
        rows = []
        for _ in range(int(rng.integers(1, 3))):
            which = str(rng.choice(["coco", "gdino"]))
            phrase = str(rng.choice(targets)) if rng.uniform() < 0.7 else str(rng.choice(distract))
            thr = 0.8 if which == "coco" else 0.4
            conf = float(np.round(thr + rng.choice([-0.25, -0.05, -0.01, 0.0, 0.01, 0.1, 0.15]), 2))
            cx, cy = int(rng.integers(20, 620)), int(rng.integers(150, 340))
            ax, ay = (int(rng.integers(3, 9)), int(rng.integers(3, 9))) if rng.uniform() < 0.15 else (int(rng.integers(15, 110)), int(rng.integers(15, 100)))
            rows.append((which, phrase, conf, (cx, cy, ax, ay), float(np.round(rng.uniform(0.08, 0.95), 2))))
        sightings[k] = rows
    return (100 + seed, steps, dataset, goal, caption), sightings


@pytest.mark.parametrize("seed", range(3))
def test_random_scripted_episodes_step_by_step_against_the_reference_policy(seed, monkeypatch):
    if GOLDEN_DIR not in sys.path:
        sys.path.insert(0, GOLDEN_DIR)
    import make_golden as mg
    import policy_script as ps

    from oracle import ref_shim
    from vlfm_amd.policy_step import ITMPolicyV2Step

    name = f"random_{seed}"
    episode, sightings = _random_script(ps, seed, steps=34)
    monkeypatch.setitem(ps.EPISODES, name, episode)
    monkeypatch.setitem(ps.SIGHTINGS, name, sightings)
    g = mg.gen_policy(name)                                          # the reference's ITMPolicyV2 over the reference's maps
    monkeypatch.setattr(golden_util, "load", lambda n: g if n == name else golden_util.load(n))
    vm, om, _, _ = ref_shim.reference_modules()
    opm, det = ref_shim.reference_object_map(), ref_shim.reference_detections()

    def make(vlm, **kw):
        obstacle = om.ObstacleMap(min_height=0.61, max_height=0.88, area_thresh=1.5, agent_radius=0.18, hole_area_thresh=100000)
        objects = opm.ObjectPointCloudMap(erosion_size=5)
        objects.reset()
        return ITMPolicyV2Step(itm=vlm.itm, coco_detector=vlm.coco, detector=vlm.gdino, sam=vlm.sam, obstacle_map=obstacle,
                               value_map=vm.ValueMap(value_channels=1, use_max_confidence=False), object_map=objects, **kw)

    pol, _ = replay_policy_episode(name, make, det.ObjectDetections, tol=0.0)
    obstacle, value, objects = pol.maps()
    assert np.array_equal(value._map, dense(g["conf_idx"], g["conf_val"], (1000, 1000), np.float32))
    assert sha(np.asarray(value._value_map, np.float64)) == str(g["value_sha"])
    assert np.array_equal(obstacle.explored_area.astype(bool), unpack_plane(g["explored"]))
    assert np.array_equal(obstacle._map.astype(bool), unpack_plane(g["obstacles"]))
    modes = [str(m) for m in g["mode"]]
    print(f"{name}: {episode[2]} / {episode[3][:20]}: {dict((m, modes.count(m)) for m in set(modes))}, model calls {len(g['calls'])}")

"""-m gpu: the FULL step of the batched harness -- scripted detector head -> class / confidence filters -> segmenter ->
ObjectPointCloudMap.update_map per surviving mask (csrc/object_cloud.hip) -> update_explored -> initialise / explore / navigate
-> goal hand-over, stop rule, batched PointNav controller -- against ``vlfm_amd.policy_step.ITMPolicyV2Step`` (the single-
environment restatement of ITMPolicyV2.act, itself pinned to the reference's own source by tests/golden/policy_*.npz) stepped
on the SAME frames, poses, cosines, detections and masks for slots 0 / 7 / 15 of a 16-environment batch (VERDICT r3 #5;
/root/reference/vlfm/policy/base_objectnav_policy.py:106-150,285-356, itm_policy.py:76-152,191-211,263-267).

Equal means equal: modes, goals, stop flags, controller resets, actions, object clouds (points AND tags: each environment's
object map owns a NumPy stream seeded like its single-environment twin), obstacle / explored planes, confidence and value maps."""
import numpy as np
import pytest
import torch

from vlfm_amd.synthetic import CAMERA_HEIGHT, HFOV_DEG, MAX_DEPTH, MIN_DEPTH

pytestmark = pytest.mark.gpu

E, SLOTS, STEPS = 16, (0, 7, 15), 64


class _Stubs:
    """The four model clients of one single-environment policy, fed from the harness's record of one slot."""

    def __init__(self, sight, env_id, target, H, W):
        self.sight, self.env_id, self.target, self.H, self.W = sight, env_id, target, H, W
        self.k = 0
        self.cos = 0.0
        self.sam_calls = 0
        outer = self

        class Itm:
            def cosine(self, image, txt):
                assert txt == f"Seems like there is a {outer.target} ahead."
                return float(outer.cos)

        class Coco:
            def predict(self, image):
                from vlfm_amd.vlm.detections import ObjectDetections

                rows = outer.sight.at(outer.env_id, outer.k, outer.target)
                boxes = torch.tensor([[(cx - ax) / W, (cy - ay) / H, (cx + ax) / W, (cy + ay) / H]
                                      for (_, _, (cx, cy, ax, ay), _) in rows], dtype=torch.float32).reshape(-1, 4)
                return ObjectDetections(boxes, torch.tensor([r[1] for r in rows], dtype=torch.float32), [r[0] for r in rows],
                                        image_source=image, fmt="xyxy")

        class Gdino:
            def predict(self, image, caption=""):
                raise AssertionError("HM3D COCO targets never reach GroundingDINO (base_objectnav_policy.py:221-233)")

        class Sam:
            def segment_bbox(self, image, bbox):
                outer.sam_calls += 1
                x0, y0, x1, y1 = [float(v) for v in bbox]
                yy, xx = np.mgrid[0:H, 0:W]
                cx, cy, ax, ay = (x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0) / 2, (y1 - y0) / 2
                return ((xx - cx) ** 2 / max(ax, 1) ** 2 + (yy - cy) ** 2 / max(ay, 1) ** 2 <= 1).astype(np.uint8)

        self.itm, self.coco, self.gdino, self.sam = Itm(), Coco(), Gdino(), Sam()


def test_full_step_harness_equals_the_single_environment_policy(gpu_device, monkeypatch):
    from vlfm_amd import policy_step
    from vlfm_amd.harness import BatchedEpisodes, ScriptedSightings
    from vlfm_amd.mapping import ObstacleMap, ValueMap
    from vlfm_amd.mapping.object_point_cloud_map import ObjectPointCloudMap
    from vlfm_amd.pointnav import WrappedPointNavResNetPolicy

    H, W = 480, 640
    torch.manual_seed(3)
    pn = WrappedPointNavResNetPolicy(None, device=gpu_device, n_envs=E, discrete_actions=True)
    # short episodes (12 turns + 4..9 search steps with distractors + 10 steps in view) so that 64 steps hold several of them,
    # every slot on the same clock as its twin (desync off)
    sight = ScriptedSightings(in_view_rate=0.7, distractor_rate=0.3, search_min=4, search_span=6, nav_steps=10, height=H, width=W,
                              desync=False)
    sim = BatchedEpisodes(E, device=gpu_device, use_blip2=False, object_maps=True, sightings=sight, select_frontiers=True,
                          pointnav=pn, world="rooms", episode_len=500)
    twins = {}
    for e in SLOTS:
        stubs = _Stubs(sight, sim.env_ids[e], sim.targets[e], H, W)
        pn1 = WrappedPointNavResNetPolicy(None, device=gpu_device, n_envs=1, discrete_actions=True)
        pn1.policy.load_state_dict(pn.policy.state_dict())
        pol = policy_step.ITMPolicyV2Step(
            camera_height=CAMERA_HEIGHT, min_depth=MIN_DEPTH, max_depth=MAX_DEPTH, camera_fov=HFOV_DEG, image_width=W,
            itm=stubs.itm, coco_detector=stubs.coco, detector=stubs.gdino, sam=stubs.sam,
            obstacle_map=ObstacleMap(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5, device=gpu_device),
            value_map=ValueMap(1, use_max_confidence=False, device=gpu_device),
            object_map=ObjectPointCloudMap(5, device=gpu_device, rng=np.random.RandomState(1000 + sim.env_ids[e])),
            pointnav=pn1)
        pol.reset(sim.targets[e])
        twins[e] = (pol, stubs)
    assert abs(twins[0][0]._fx - sim.fx) == 0.0 and twins[0][0]._camera_fov == sim.fov
    current_tf = {}
    # the harness's poses come from the tour's exact (cos, sin) table; hand the twin the same matrix instead of cos(yaw)
    monkeypatch.setattr(policy_step, "xyz_yaw_to_tf_matrix", lambda xyz, yaw: current_tf["tf"])
    seen = {"navigate": 0, "explore": 0, "initialize": 0, "stop": 0, "sam": 0, "reset": 0, "episodes": 0}
    for k in range(STEPS):
        t = sim.t % sim.episode_len
        depth = sim.rooms.frame(t).cpu().numpy()
        rgb = sim.rgb_pool[t % sim.rgb_pool.shape[0]].cpu().numpy()
        poses, tf = sim.pose_table[t], sim.tf_table[t]
        sim.step()
        torch.cuda.synchronize()
        cos = sim.last_cosines.double().reshape(-1).cpu().numpy()
        acts = sim.last_actions.cpu().numpy().reshape(-1)
        for e in SLOTS:
            pol, stubs = twins[e]
            stubs.k, stubs.cos = k, cos[e]
            current_tf["tf"] = tf[e]
            r = pol.step(rgb[e], depth[e], float(poses[e, 0]), float(poses[e, 1]), float(poses[e, 2]))
            where = (k, e, r.mode)
            assert sim.last_modes[e] == r.mode, where
            if r.goal is None:
                assert np.isnan(sim.last_goals[e]).all(), where
            else:
                assert np.array_equal(sim.last_goals[e], np.asarray(r.goal, np.float64)), (where, sim.last_goals[e], r.goal)
                assert abs(sim.last_rho_theta[e, 0] - r.rho) <= 1e-12 and abs(sim.last_rho_theta[e, 1] - r.theta) <= 1e-12, where
                assert bool(sim.last_resets[e]) == bool(r.pointnav_reset), where
            assert bool(sim.last_stops[e]) == bool(r.stop), where
            assert int(acts[e]) == int(r.action), (where, int(acts[e]), r.action)
            seen[r.mode] += 1
            seen["stop"] += int(r.stop)
            seen["reset"] += int(r.pointnav_reset)
            got, want = sim.object_maps[e].clouds, pol.maps()[2].clouds
            if sim.last_episode_end[e]:          # the script says the robot arrived: both sides start the next episode in place
                assert sight.episode_ends(sim.env_ids[e], k) and not got      # (the harness has reset this slot at the end of its step)
                pol.reset(sim.targets[e])
                seen["episodes"] += 1
                continue
            assert sorted(got) == sorted(want), where
            for name in want:
                assert np.array_equal(got[name], want[name]), (where, name, got[name].shape, want[name].shape)
    sim.check()
    # the script exercised every branch: object goals, frontier goals, stops, filtered distractors, SAM calls
    seen["sam"] = sum(twins[e][1].sam_calls for e in SLOTS)
    assert seen["initialize"] >= 24 * len(SLOTS) and seen["explore"] > 0 and seen["navigate"] > 0 and seen["sam"] > 0, seen
    assert seen["reset"] > 0 and seen["episodes"] >= 2 * len(SLOTS), seen
    assert sim.object_stats["detections"] > 0 and sim.object_stats["cloud_updates"] > 0, sim.object_stats
    # maps at the end: the batch slot equals its single-environment twin, bit for bit
    obst = sim.obstacles._unpack(sim.obstacles.obstacle_bits).cpu().numpy().astype(bool)
    expl = sim.obstacles.explored.cpu().numpy().astype(bool)
    conf, value = sim.values.conf.cpu().numpy(), sim.values.value.cpu().numpy()
    for e in SLOTS:
        om, vm, _ = twins[e][0].maps()
        assert np.array_equal(obst[e], np.asarray(om._map).astype(bool)), e
        assert np.array_equal(expl[e], np.asarray(om.explored_area).astype(bool)), e
        assert np.array_equal(conf[e], vm._map), e
        assert np.array_equal(value[e].reshape(vm._value_map.shape), vm._value_map), e


def test_blip2_beside_the_detector_equals_blip2_behind_it(gpu_device):
    """At small batches the harness enqueues the BLIP-2 forward on its own stream BEFORE the detector (harness.vlm_stream) so that
    the two run beside each other.  The order is not allowed to change anything: same cosines, same maps, same detections and
    actions as with the forward behind the detector on the main stream (concurrent_vlm_max_envs=0), step by step."""
    from vlfm_amd.harness import BatchedEpisodes, ScriptedSightings
    from vlfm_amd.pointnav import WrappedPointNavResNetPolicy
    from vlfm_amd.vlm.blip2itm import BLIP2ITM
    from vlfm_amd.vlm.sam import MobileSAM
    from vlfm_amd.vlm.yolov7 import YOLOv7

    n = 4
    blip2 = BLIP2ITM(device=gpu_device, allow_random_init=True)
    det = YOLOv7(device=gpu_device, allow_random_init=True, width_multiple=0.25)
    sam = MobileSAM(device=gpu_device, allow_random_init=True)
    runs = []
    for limit in (16, 0):
        torch.manual_seed(5)
        sight = ScriptedSightings(in_view_rate=0.9, distractor_rate=0.2, search_min=2, search_span=3, nav_steps=8, height=480,
                                  width=640, desync=False)
        sim = BatchedEpisodes(n, device=gpu_device, blip2=blip2, detector=det, sam=sam, select_frontiers=True, object_maps=True,
                              sightings=sight, scripted_masks=True, concurrent_vlm_max_envs=limit,
                              pointnav=WrappedPointNavResNetPolicy(None, device=gpu_device, n_envs=n, discrete_actions=True))
        assert (sim.vlm_stream is not None) == (limit > 0)
        trace = []
        for _ in range(22):
            sim.step()
            torch.cuda.synchronize()
            trace.append((sim.last_cosines.double().cpu().numpy().copy(), list(sim.last_modes),
                          None if sim.last_goals is None else np.array(sim.last_goals, copy=True),
                          [int(d.num_detections) for d in sim.last_detections]))
        sim.check()
        runs.append((trace, sim.values.conf.cpu().numpy().copy(), sim.values.value.cpu().numpy().copy(),
                     sim.obstacles.explored.cpu().numpy().copy()))
        del sim
    (ta, ca, va, ea), (tb, cb, vb, eb) = runs
    for k, (a, b) in enumerate(zip(ta, tb)):
        assert np.array_equal(a[0], b[0]), k
        assert a[1] == b[1] and a[3] == b[3], k
        assert (a[2] is None and b[2] is None) or np.array_equal(a[2], b[2], equal_nan=True), k
    assert np.array_equal(ca, cb) and np.array_equal(va, vb) and np.array_equal(ea, eb)
    assert any(m == "navigate" for a in ta for m in a[1])      # the object-map branch ran


def test_scripted_head_speaks_through_the_real_nms(gpu_device):
    """in the timed full step the YOLOv7 network runs on every frame but its random logits decide nothing.  The
    scripted head's candidates (24 jittered boxes per sighting, the scripted confidence on the best one) are now written into the
    network's raw prediction, and the detector's own non_max_suppression / scale_coords / rounding (yolov7.py:91-110) turn them into
    the detections: one per sighting, the scripted class and confidence, the scripted box within the rounding of the path."""
    from vlfm_amd.harness import BatchedEpisodes, ScriptedSightings
    from vlfm_amd.vlm import det_ops
    from vlfm_amd.vlm.yolov7 import YOLOv7

    n = 16
    det = YOLOv7(device=gpu_device, allow_random_init=True, width_multiple=0.25)
    sight = ScriptedSightings(in_view_rate=0.9, distractor_rate=0.5, search_min=2, search_span=3, nav_steps=8, height=480, width=640)
    sim = BatchedEpisodes(n, device=gpu_device, use_blip2=False, detector=det, object_maps=True, sightings=sight, scripted_masks=True)
    assert sim.scripted_through_nms
    det_ops.NMS_STATS.update(frames=0, candidates=0, boxes_in=0)
    seen = 0
    for _ in range(30):
        t_ep = sim.t % sim.episode_len
        want = sim._scripted_detections(t_ep)
        sim.step()
        torch.cuda.synchronize()
        got = sim.last_detections
        for e in range(n):
            # (the harness filtered `got` by class and confidence in place: compare what survives both)
            w = want[e]
            w.filter_by_class(sim.targets[e].split("|"))
            w.filter_by_conf(sim.det_threshold)
            assert got[e].num_detections == w.num_detections, (sim.t, e, got[e].phrases, w.phrases)
            for i in range(w.num_detections):
                assert got[e].phrases[i] == w.phrases[i]
                assert abs(float(got[e].logits[i]) - float(w.logits[i])) <= 2e-3
                px = (got[e].boxes[i] - w.boxes[i]).abs() * torch.tensor([640.0, 480.0, 640.0, 480.0])
                assert float(px.max()) <= 2.0, (px, got[e].boxes[i], w.boxes[i])
                seen += 1
    assert seen > 10 and sim.object_stats.get("head_mismatch", 0) == 0
    assert det_ops.NMS_STATS["candidates"] >= 24 * seen          # the NMS really had the clusters to suppress

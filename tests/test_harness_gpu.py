"""-m gpu: parity of THE OBJECT bench.py TIMES -- ``BatchedEpisodes.step()`` with its device-side renderer, its two-stream
choreography (depth ingest + obstacle pipeline on a side stream, BLIP-2 on the main stream, value-map fusion joining both,
the column-max key buffer handed back zeroed by the fused kernel's last workgroup), ``fast_forward()`` and device-resident
cosines from ``cosine_batch`` / ``cosine_batch_graphed`` -- against the oracle (oracle/ref_obstacle_map.py +
oracle/ref_value_map.py = the reference's ObstacleMap.update_map / ValueMap.update_map, obstacle_map.py:55-153,
value_map.py:100-128).

Per step the frame the harness rendered on the device (``rooms.frame(t)``) and the cosines it fed to the fusion are
downloaded and given to the oracle for slots 0 / 7 / 15; then obstacle / navigable / explored planes, frontier pixels, the
f32 confidence map, the f64 value map and the frontier medians must all be EQUAL (the device value map is f64 with the
reference's own promotion rule, so no tolerance is needed).  A race between the key hand-back and the next step's
atomicMax ingest, a missing stream dependency or a renderer that differs from the host renderer would show up here."""
import copy

import numpy as np
import pytest
import torch

from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, camera_intrinsics, depth_from_profile, integrate, plan_actions, \
    wall_profile

pytestmark = pytest.mark.gpu

E = 16
SLOTS = (0, 7, 15)
KW = dict(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5)


def test_device_renderer_equals_host_renderer_bit_for_bit(gpu_device):
    """RoomsRenderer.render(t) (torch f64 on the device) == synthetic.depth_from_profile(wall_profile(...)) (NumPy f64 on
    the host), for every environment of a batch at steps spread over the tour."""
    from vlfm_amd.harness import RoomsRenderer

    L = 500
    env_ids = list(range(E))
    rr = RoomsRenderer(env_ids, L, 480, 640, gpu_device)
    poses = integrate(plan_actions(2 * L))
    for t in (0, 5, 13, 77, 150, 209, 333, 499):
        got = rr.render(t).cpu().numpy()
        assert got.dtype == np.float32 and got.shape == (E, 480, 640)
        for e in env_ids:
            x, y, k = poses[(37 * e) % L + t]
            want = depth_from_profile(wall_profile(x, y, k, 640), 480)
            assert np.array_equal(got[e], want), (t, e, np.abs(got[e] - want).max())
    rr.prepare(150, 4)   # the pre-rendered window bench.py uses serves the same frames
    assert torch.equal(rr.frame(151), rr.render(151)) and torch.equal(rr.frame(160), rr.render(160))


class _Follower:
    """The oracle side of one harness: reference obstacle maps (shared between harnesses that see the same frames) and
    reference value maps for the watched slots."""

    def __init__(self, oms, vms):
        self.oms, self.vms = oms, vms

    def step_values(self, depth, cos, tf, fov):
        for e in SLOTS:
            self.vms[e].update_map(np.array([cos[e]]), depth[e].copy(), tf[e], MIN_DEPTH, MAX_DEPTH, fov)

    def compare(self, sim, where):
        obst = sim.obstacles._unpack(sim.obstacles.obstacle_bits).cpu().numpy().astype(bool)
        nav = sim.obstacles._unpack(sim.obstacles.navigable_bits).cpu().numpy().astype(bool)
        expl = sim.obstacles.explored.cpu().numpy().astype(bool)
        fr = sim.obstacles.frontiers_px()
        conf, value = sim.values.conf.cpu().numpy(), sim.values.value.cpu().numpy()
        wps, env_of = sim.obstacles.frontier_list()
        for e in SLOTS:
            om, vm = self.oms[e], self.vms[e]
            assert np.array_equal(obst[e], np.asarray(om._map).astype(bool)), (where, e, "obstacle plane")
            assert np.array_equal(nav[e], np.asarray(om._navigable_map).astype(bool)), (where, e, "navigable plane")
            assert np.array_equal(expl[e], np.asarray(om.explored_area).astype(bool)), (where, e, "explored plane")
            want_px = np.asarray(om._frontiers_px, np.float64).reshape(-1, 2)
            assert np.array_equal(fr[e].reshape(-1, 2), want_px), (where, e, "frontier pixels")
            assert np.array_equal(conf[e], vm._map), (where, e, "confidence map", np.abs(conf[e] - vm._map).max())
            assert np.array_equal(value[e], vm._value_map), (where, e, "value map", np.abs(value[e] - vm._value_map).max())
            if len(want_px):
                # what step() scored: the disc medians of this step's frontiers on the map AFTER this step's fusion
                got = np.asarray(sim.last_frontier_values, np.float64).reshape(-1)[env_of == e]
                pts = wps[env_of == e]
                assert np.array_equal(pts, np.asarray(om.frontiers, np.float64).reshape(-1, 2)), (where, e)
                s_wp, s_val = vm.sort_waypoints(pts, 0.5)
                order = np.argsort([-v for v in got])                      # value_map.py:183 on the harness's medians
                assert np.array_equal(np.asarray(s_val, np.float64), got[order]), (where, e, "frontier medians")
                assert np.array_equal(np.asarray(s_wp), pts[order]), (where, e, "sort_waypoints permutation")


def test_timed_step_two_streams_fast_forward_and_blip2_against_the_oracle(gpu_device):
    from oracle.ref_obstacle_map import RefObstacleMap
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.harness import BatchedEpisodes
    from vlfm_amd.vlm.blip2itm import BLIP2ITM

    fx, fy, fov = camera_intrinsics(640)
    FF, STEPS = 150, 60
    torch.manual_seed(0)
    blip2 = BLIP2ITM(device=gpu_device, allow_random_init=True)
    common = dict(device=gpu_device, world="rooms", episode_len=500)
    sims = {
        "two streams, stub cosines": BatchedEpisodes(E, use_blip2=False, overlap=True, **common),
        "one stream, stub cosines": BatchedEpisodes(E, use_blip2=False, overlap=False, **common),
        "two streams, BLIP-2 cosine_batch": BatchedEpisodes(E, blip2=blip2, overlap=True, graph_blip2=False, **common),
        "two streams, BLIP-2 from a HIP graph": BatchedEpisodes(E, blip2=blip2, overlap=True, graph_blip2=True, **common),
    }
    names = list(sims)
    oms = {e: RefObstacleMap(**KW) for e in SLOTS}
    stub = _Follower(oms, {e: RefValueMap(1, use_max_confidence=False) for e in SLOTS})
    followers = {}

    def advance(fast: bool, where: str):
        """One step of every harness; the frames are the same for all of them (asserted), so the reference obstacle maps
        advance once; value maps advance per follower."""
        t = sims[names[0]].t % 500
        depth = sims[names[0]].rooms.frame(t).cpu().numpy()
        tf = sims[names[0]].tf_table[t]
        for e in SLOTS:
            oms[e].update_map(depth[e].copy(), tf[e], MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
        fed = set()
        for n in names:
            sim = sims[n]
            assert sim.t % 500 == t and torch.equal(sim.rooms.frame(t), sims[names[0]].rooms.frame(t))
            if fast:
                sim.fast_forward(1)
            else:
                sim.step()
            torch.cuda.synchronize()
            cos = sim.last_cosines.double().reshape(-1).cpu().numpy()
            f = followers.get(n, stub)
            if id(f) not in fed:
                f.step_values(depth, cos, tf, fov)
                f.last_cos = cos
                fed.add(id(f))
            assert np.array_equal(cos, f.last_cos), (where, n, "harnesses sharing a follower must feed the same cosines")
        return t

    for i in range(FF):
        advance(True, f"fast-forward step {i}")
        if i % 25 == 24 or i == FF - 1:
            for n in names:
                stub.compare(sims[n], f"{n}: fast-forward step {i}")
    # from here on the BLIP-2 harnesses feed their own (device-resident) cosines: each gets its own reference value maps
    for n in names[2:]:
        followers[n] = _Follower(oms, {e: copy.deepcopy(stub.vms[e]) for e in SLOTS})
    seen_real = []
    for i in range(STEPS):
        t = advance(False, f"step {FF + i}")
        for n in names:
            followers.get(n, stub).compare(sims[n], f"{n}: step {FF + i} (episode step {t})")
        seen_real.append(followers[names[2]].last_cos.copy())
    for n in names:
        sims[n].check()
    # the BLIP-2 harnesses really went through the network: cosines of a random-init model are not the stub's U(0.15, 0.45)
    real = np.stack(seen_real)
    assert np.isfinite(real).all() and not np.array_equal(real[-1], stub.last_cos)
    # eager forward and graph replay of the same network on the same frames agree to f16 noise
    g = followers[names[3]].last_cos
    assert np.abs(g - real[-1]).max() <= 5e-3, np.abs(g - real[-1]).max()

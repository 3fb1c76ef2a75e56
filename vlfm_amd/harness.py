"""Batched-episode harness: E independent synthetic ObjectNav episodes resident on ONE GPU, stepped together.

Replaces the reference's single-env Habitat eval loop (vlfm/utils/vlfm_trainer.py:164-174, which raises when
distributed, :65-66) for throughput measurement.  One `step()` performs, for every resident environment, the
per-step hot path of ITMPolicyV2 (vlfm/policy/itm_policy.py:251-261):

    _cache_observations -> ObstacleMap.update_map      (habitat_policies.py:193-201)      [depth ingest + obstacle kernels]
    _update_value_map   -> BLIP2ITMClient.cosine        (itm_policy.py:191-203)            [batched in-process BLIP-2 ITC]
                        -> ValueMap.update_map          (itm_policy.py:204-206)            [one launch for all envs]
    _explore            -> ValueMap.sort_waypoints      (itm_policy.py:263-267)            [disc medians per frontier]

and, with a detector / segmenter / ``object_maps=True`` (the configs[2] "full" step, base_objectnav_policy.py:106-150,285-356):

    _update_object_map  -> detector.predict (YOLOv7 | GroundingDINO), class + confidence filters   (:221-241)
                        -> MobileSAM.segment_bbox per surviving box                               (:311-321)
                        -> ObjectPointCloudMap.update_map per mask, update_explored per step      (:337-350)
    act                 -> initialise (12 x TURN_LEFT) | explore (best frontier) | navigate (object goal, stop rule)
                        -> PointNav controller on the chosen goal                                  (:126-135,243-283)

Episodes are independent, so multi-GPU scaling is pure sharding (contiguous blocks: env e -> rank e // envs_per_rank,
vlfm_amd/distributed.py); the only collective is the metric all-reduce in bench.py.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .mapping.base_map import require_gpu
from .mapping.value_map import ValueMapBatch
from .synthetic import BOXES, CAMERA_HEIGHT, HEADINGS, MAX_DEPTH, MIN_DEPTH, YAWS, Trajectory, camera_intrinsics, \
    depth_frame, integrate, plan_actions, pose_to_tf, rgb_frame, tf_of

PROMPT = "Seems like there is a target_object ahead."  # vlfm/policy/base_objectnav_policy.py:377
TARGETS = ["chair", "bed", "potted plant", "toilet", "tv", "couch"]  # HM3D ObjectNav categories


class ScriptedSightings:
    """A deterministic detector HEAD and episode script for networks without pretrained weights: which (environment, step) pairs
    carry a detection, of what, how confident, where in the image and how far away -- a pure function of (env_id, step), so that
    the workload of the stages behind the detector (MobileSAM, ObjectPointCloudMap) is STATED instead of being decided by random
    logits.  An environment lives through ObjectNav episodes shaped like the reference's: 12 initialisation turns, a SEARCH phase of
    ``search_min + (hash mod search_span)`` steps in which only distractors show up (``distractor_rate`` of the steps: a wrong class
    or a low-confidence target, which the filters of base_objectnav_policy.py:231-233 must drop before SAM), then the target IN VIEW
    for ``nav_steps`` steps (``in_view_rate`` of them carry a confidence-0.9 detection: survives both the 0.8 YOLOv7 and the 0.4
    GroundingDINO threshold) while the policy navigates to it; after the last of these the robot "arrives": the episode ends like
    the reference's does on STOP, and the environment starts the next one in place (maps, object map, selector and controller
    reset).  ``desync``: environment e starts ``hash(e)`` steps into its first episode, so a batch is spread over all phases.
    The object is an ellipse painted into the depth frame (nearer surfaces win), so the object cloud is an object, not a wall."""

    def __init__(self, in_view_rate: float = 0.8, distractor_rate: float = 0.0625, search_min: int = 60, search_span: int = 120,
                 nav_steps: int = 30, height: int = 480, width: int = 640, desync: bool = True) -> None:
        self.in_view_rate, self.distractor_rate = in_view_rate, distractor_rate
        self.search_min, self.search_span, self.nav_steps = search_min, search_span, nav_steps
        self.H, self.W, self.desync = height, width, desync

    @staticmethod
    def _mix(a: int, b: int, salt: int) -> int:
        h = (a * 2654435761 + b * 40503 + salt * 97 + 0x9E3779B9) & 0xFFFFFFFF
        h ^= h >> 16
        h = (h * 0x85EBCA6B) & 0xFFFFFFFF
        h ^= h >> 13
        h = (h * 0xC2B2AE35) & 0xFFFFFFFF
        return h ^ (h >> 16)

    def episode_length(self, env_id: int, k: int) -> int:
        return 12 + self.search_min + self._mix(env_id, k, 3) % max(self.search_span, 1) + self.nav_steps

    def locate(self, env_id: int, step: int):
        """(episode index, step inside the episode, episode length) of environment ``env_id`` at harness step ``step``."""
        g = step + (self._mix(env_id, 0, 9) % self.episode_length(env_id, 0) if self.desync else 0)
        k = 0
        while g >= self.episode_length(env_id, k):
            g -= self.episode_length(env_id, k)
            k += 1
        return k, g, self.episode_length(env_id, k)

    def episode_ends(self, env_id: int, step: int) -> bool:
        _, s_, n = self.locate(env_id, step)
        return s_ == n - 1

    def mean_detections_per_env_step(self) -> float:
        return self.in_view_rate * self.nav_steps / (12 + self.search_min + (self.search_span - 1) / 2.0 + self.nav_steps)

    def at(self, env_id: int, step: int, target: str):
        """[(phrase, confidence, (cx, cy, ax, ay) pixels, normalised depth)] for one environment-step."""
        _, s_, n = self.locate(env_id, step)
        if s_ < 12:
            return []
        in_view = s_ >= n - self.nav_steps
        u = self._mix(env_id, step, 1) / 2.0 ** 32
        if u >= (self.in_view_rate if in_view else self.distractor_rate):
            return []
        g = self._mix(env_id, step, 2)
        cx = int(self.W * (0.2 + 0.6 * ((g & 0xFF) / 255.0)))
        cy = int(self.H * (0.45 + 0.15 * (((g >> 8) & 0xFF) / 255.0)))
        ax, ay = 40 + ((g >> 16) & 0x1F), 50 + ((g >> 21) & 0x1F)
        depth = 0.3 + 0.3 * (((g >> 26) & 0x3F) / 63.0)
        if in_view:
            return [(target, 0.9, (cx, cy, ax, ay), depth)]
        wrong = "tv" if target != "tv" else "chair"
        return [(wrong, 0.93, (cx, cy, ax, ay), depth)] if (g & 1) else [(target, 0.35, (cx, cy, ax, ay), depth)]


SAM_BATCH_BUCKETS = (1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256)


def sam_batch_bucket(n: int) -> int:
    """The batch size the segmenter runs ``n`` boxes at: the next of a fixed set of sizes (<= 25 % padding from 4 boxes on)."""
    for b in SAM_BATCH_BUCKETS:
        if b >= n:
            return b
    return -(-n // 64) * 64


def ellipse_masks(ellipses, height: int, width: int, device) -> torch.Tensor:
    """[n,H,W] bool: (x - cx)^2 / max(ax, 1)^2 + (y - cy)^2 / max(ay, 1)^2 <= 1 in f64 for ``ellipses`` [n,4] = (cx, cy, ax, ay)."""
    e = torch.as_tensor(np.asarray(ellipses, np.float64).reshape(-1, 4), device=device)
    yy = torch.arange(height, device=device, dtype=torch.float64)[None, :, None]
    xx = torch.arange(width, device=device, dtype=torch.float64)[None, None, :]
    cx, cy = e[:, 0, None, None], e[:, 1, None, None]
    ax, ay = e[:, 2, None, None].clamp(min=1.0), e[:, 3, None, None].clamp(min=1.0)
    return (xx - cx) ** 2 / ax ** 2 + (yy - cy) ** 2 / ay ** 2 <= 1


class RoomsRenderer:
    """Depth frames of the consistent rooms-and-pillars world (vlfm_amd/synthetic.py) for E environments at once, ray-cast ON
    THE DEVICE (f64, the arithmetic of synthetic.wall_profile / depth_from_profile batched over environments): environment
    e walks the planned tour starting ``37 * env_id mod L`` steps in, so the batch sees rooms, doorways, pillars and a dozen
    or more simultaneous frontiers instead of the per-frame random walls of depth_frame().  ``prepare(t0, n)`` renders a
    window of steps ahead of a timed region (inputs resident in HBM before the clock starts); other steps render on the fly."""

    def __init__(self, env_ids, episode_len: int, height: int, width: int, device) -> None:
        self.L, self.H, self.W, self.device = episode_len, height, width, device
        poses = integrate(plan_actions(2 * episode_len))            # two laps of the tour: no wrap inside an episode
        offs = [(37 * int(i)) % episode_len for i in env_ids]
        at = [[poses[o + t] for o in offs] for t in range(episode_len)]
        self.pose_table = np.array([[(x, y, YAWS[k]) for (x, y, k) in row] for row in at])       # [L,E,3]
        self.tf_table = np.stack([np.stack([tf_of(x, y, k) for (x, y, k) in row]) for row in at])  # [L,E,4,4]
        f64 = dict(dtype=torch.float64, device=device)
        self.xy = torch.tensor([[(x, y) for (x, y, _) in row] for row in at], **f64)             # [L,E,2]
        self.cs = torch.tensor([[HEADINGS[k] for (_, _, k) in row] for row in at], **f64)        # [L,E,2] exact (cos, sin)
        fx, fy, _ = camera_intrinsics(width)
        self.m = (-(torch.arange(width, **f64) - width // 2) / fx)[None, :, None]               # [1,W,1]
        rows = torch.arange(height, **f64) - height // 2
        self.floor = torch.where(rows > 0, CAMERA_HEIGHT * fy / rows.clamp(min=1e-9),
                                 torch.full_like(rows, float("inf")))[None, :, None]            # [1,H,1]
        self.boxes = torch.tensor(BOXES, **f64)                                                  # [B,4]
        self.window = None
        self.window_t0 = 0
        self.painter = None      # optional (t, frames [E,H,W]) -> frames: scripted objects in front of the walls

    @torch.no_grad()
    def render(self, t: int) -> torch.Tensor:
        """[E,H,W] f32 normalised depth of episode step t."""
        d = self._render_walls(t)
        return self.painter(t, d) if self.painter is not None else d

    def _render_walls(self, t: int) -> torch.Tensor:
        x, y = self.xy[t, :, 0][:, None, None], self.xy[t, :, 1][:, None, None]
        c, s = self.cs[t, :, 0][:, None, None], self.cs[t, :, 1][:, None, None]
        dx, dy = c - s * self.m, s + c * self.m                                                  # [E,W,1]
        tiny = 1e-12
        dx = torch.where(dx.abs() < tiny, torch.full_like(dx, tiny), dx)
        dy = torch.where(dy.abs() < tiny, torch.full_like(dy, tiny), dy)
        b = self.boxes
        tx0, tx1 = (b[:, 0] - x) / dx, (b[:, 2] - x) / dx                                        # [E,W,B]
        ty0, ty1 = (b[:, 1] - y) / dy, (b[:, 3] - y) / dy
        tmin = torch.maximum(torch.minimum(tx0, tx1), torch.minimum(ty0, ty1))
        tmax = torch.minimum(torch.maximum(tx0, tx1), torch.maximum(ty0, ty1))
        hit = (tmax >= tmin.clamp(min=0.0)) & (tmin > 0.0)
        wall = torch.where(hit, tmin, torch.full_like(tmin, float("inf"))).amin(dim=2).float().double()   # f32 like the host path
        d = torch.minimum(wall[:, None, :], self.floor)                                          # [E,H,W]
        return ((d - MIN_DEPTH) / (MAX_DEPTH - MIN_DEPTH)).clamp(1e-3, 1.0).float()

    def prepare(self, t0: int, n: int) -> None:
        self.window = torch.stack([self.render((t0 + i) % self.L) for i in range(n)])
        self.window_t0 = t0

    def frame(self, t: int) -> torch.Tensor:
        if self.window is not None and 0 <= t - self.window_t0 < self.window.shape[0]:
            return self.window[t - self.window_t0]
        return self.render(t)


class BatchedEpisodes:
    def __init__(self, n_envs: int, device=None, height: int = 480, width: int = 640, env_offset: int = 0,
                 blip2=None, use_blip2: bool = True, frame_pool: int = 4, map_size: int = 1000,
                 n_frontiers: int = 8, sync_explored: bool = False, obstacle: bool = True,
                 episode_len: int = 500, overlap: bool = True, detector=None, sam=None, sam_every: int = 4,
                 graph_blip2: Optional[bool] = None, host_inputs: bool = False, select_frontiers: bool = False,
                 pointnav=None, world: str = "rooms", object_maps: bool = False,
                 sightings: Optional["ScriptedSightings"] = None, scripted_masks: bool = False,
                 coco_threshold: float = 0.8, non_coco_threshold: float = 0.4, pointnav_stop_radius: float = 0.9,
                 object_map_erosion_size: float = 5, concurrent_vlm_max_envs: int = 0) -> None:
        self.device = require_gpu(device)
        # a rank waiting for its GPU must not hold a host core (bench.py `host`).  Effective only before the device's first
        # stream exists (bench.py sets it first thing); here it is best effort: a warning on failure, VLFM_HOST_WAIT=spin opts out
        _lib.try_host_wait_blocking(self.device)
        self.E, self.H, self.W, self.S = n_envs, height, width, map_size
        self.fx, self.fy, self.fov = camera_intrinsics(width)
        self.episode_len = episode_len
        self.env_ids = [env_offset + e for e in range(n_envs)]
        self.targets = [TARGETS[i % len(TARGETS)] for i in self.env_ids]
        self.prompts = [PROMPT.replace("target_object", t.replace("|", "/")) for t in self.targets]
        self.values = ValueMapBatch(n_envs, 1, map_size, use_max_confidence=False, device=self.device)
        self.n_frontiers = n_frontiers
        self.t = 0
        self.episodes_done = 0
        # synthetic observations live in HBM before the timed region starts (bench contract): a small pool of
        # distinct frames per env, cycled; the scripted poses of a whole episode are tabulated up front as well
        rng = np.random.Generator(np.random.PCG64(99991 + env_offset))
        # world "rooms": every environment walks the consistent rooms-and-pillars world (frames ray-cast on the device,
        # RoomsRenderer); world "random": SURVEY 8d's per-frame random wall profiles on scripted random-walk poses
        assert world in ("rooms", "random")
        self.rooms = RoomsRenderer(self.env_ids, episode_len, height, width, self.device) \
            if world == "rooms" and not host_inputs else None
        depth_pool = torch.from_numpy(np.stack([
            np.stack([depth_frame(rng, height, width) for _ in range(n_envs)])
            for _ in range(1 if self.rooms is not None else frame_pool)]))
        rgb_pool = torch.from_numpy(np.stack([
            np.stack([rgb_frame(rng, height, width) for _ in range(n_envs)]) for _ in range(frame_pool)]))
        # host_inputs: the simulator hands over HOST buffers every step (what the reference's API receives); the frames
        # then cross PCIe inside the step -- the "PCIe-inclusive" rate of DESIGN.md, never the headline value
        self.host_inputs = host_inputs
        if host_inputs:
            self.depth_pool, self.rgb_pool = depth_pool.pin_memory(), rgb_pool.pin_memory()
            self.depth_dev = torch.empty(depth_pool.shape[1:], dtype=depth_pool.dtype, device=self.device)
            self.rgb_dev = torch.empty(rgb_pool.shape[1:], dtype=rgb_pool.dtype, device=self.device)
        else:
            self.depth_pool, self.rgb_pool = depth_pool.to(self.device), rgb_pool.to(self.device)
        if self.rooms is not None:
            self.pose_table, self.tf_table = self.rooms.pose_table, self.rooms.tf_table
        else:
            trajs = [Trajectory(i) for i in self.env_ids]
            self.pose_table = np.array([[tr.step() for tr in trajs] for _ in range(episode_len)])  # [L,E,3]
            self.tf_table = np.stack([np.stack([pose_to_tf(x, y, yaw) for (x, y, yaw) in row]) for row in self.pose_table])
        self.blip2 = blip2
        if use_blip2 and blip2 is None:
            from .vlm.blip2itm import BLIP2ITM

            self.blip2 = BLIP2ITM(device=self.device, allow_random_init=True)  # synthetic-episode harness: throughput only
        # small batches are launch-bound: replay the BLIP-2 forward from a captured HIP graph
        self.graph_blip2 = (n_envs <= 4) if graph_blip2 is None else graph_blip2  # measured: +43 % at 1 env, none at 8
        self.stub_rng = np.random.Generator(np.random.PCG64(7 + env_offset))
        self.obstacles = None
        if obstacle:
            from .mapping.obstacle_map import ObstacleMapBatch

            self.obstacles = ObstacleMapBatch(n_envs, min_height=0.61, max_height=0.88, agent_radius=0.18,
                                              area_thresh=1.5, size=map_size, device=self.device)
            if sync_explored:
                self.values.explored_bits = self.obstacles.explored_bits
        # The obstacle pipeline is a handful of latency-bound workgroups per environment: it runs on its own HIP
        # stream beside the BLIP-2 GEMMs (which fill the chip) instead of in front of them.
        # "full" ITMPolicyV2 step (configs[2]): object detector on every frame (YOLOv7 for the COCO targets of HM3D,
        # base_objectnav_policy.py:221-233) and MobileSAM on the boxes that survive (:311-321).  With random-init networks
        # detections carry no meaning, so SAM is exercised on one synthetic box for every ``sam_every``-th environment-step.
        self.detector, self.sam, self.sam_every = detector, sam, sam_every
        self.detector_is_prompted = detector is not None and "caption" in getattr(detector, "__dict__", {})
        self.gdino_caption = " . ".join(TARGETS) + " ."
        self.last_detections = None
        self.last_masks = None
        # the stage BEHIND the detector (base_objectnav_policy.py:311-350): one ObjectPointCloudMap per environment, each with
        # its own NumPy stream (the reference draws from the global generator; E interleaved episodes need E streams)
        self.object_maps = None
        if object_maps:
            from .mapping.object_point_cloud_map import ObjectPointCloudMap

            self.object_maps = [ObjectPointCloudMap(object_map_erosion_size, device=self.device,
                                                    rng=np.random.RandomState(1000 + i)) for i in self.env_ids]
        self.sightings, self.scripted_masks = sightings, scripted_masks
        self.scripted_through_nms = True    # YOLOv7: the scripted head's candidates go through the detector's real post-processing
        self._sight_cache: Dict[int, List] = {}
        self._sched_cache: Dict[int, tuple] = {}
        if sightings is not None and self.rooms is not None:
            self.rooms.painter = self._paint_sightings
        self.det_threshold = non_coco_threshold if self.detector_is_prompted else coco_threshold   # :231-233
        self.stop_radius = pointnav_stop_radius
        self.last_modes: List[str] = []
        self.last_episode_end = np.zeros(n_envs, bool)
        self.last_stops = np.zeros(n_envs, bool)
        self.last_resets = np.zeros(n_envs, bool)
        self.last_rho_theta = np.full((n_envs, 2), np.nan)
        self.object_stats = {"detections": 0, "masks": 0, "cloud_updates": 0, "env_steps": 0,
                             "modes": {"initialize": 0, "explore": 0, "navigate": 0}}
        # SAM + ObjectPointCloudMap updates (a chain of small kernels and host read-backs) run beside the BLIP-2 forward (step()).
        # In the full step from 128 environments on, their stream and the map stream are HIGH-PRIORITY queues: beside a forward whose GEMMs hold every CU for a millisecond
        # at a time, the chain's small kernels otherwise wait their turn and the step ends when THEY do -- measured on one box, three
        # runs each (profiles/r05_side_stream_priority.txt): 726-738 env-steps/s at priority 0 (other boxes: 845-849, i.e. the slow
        # mode is box- or run-dependent) against 819-828 at priority -1; at 64 environments the priority costs 7 % (746-767 -> 702-715:
        # there the forward is the shorter part and is the one being pushed aside), so it is not set below 128.
        # VLFM_SIDE_PRIORITY overrides (diagnostic).
        full_step = detector is not None and object_maps
        prio = int(os.environ["VLFM_SIDE_PRIORITY"]) if "VLFM_SIDE_PRIORITY" in os.environ else (-1 if n_envs >= 128 and full_step else 0)
        self.map_stream = torch.cuda.Stream(self.device, priority=prio) if overlap else None
        self.obj_stream = torch.cuda.Stream(self.device, priority=prio) if overlap else None
        # (the object stream alone at priority -1: 790-796; both: 819-828; the headline -- no detector -- keeps priority 0: -0.3 % with it)
        # Optional (concurrent_vlm_max_envs > 0, VLFM_VLM_BESIDE): at small batches neither the detector (a HIP graph of ~1 700 short
        # kernels for GroundingDINO) nor the BLIP-2 forward of 8 frames fills the chip, so the BLIP-2 forward can be enqueued FIRST,
        # on its own stream, with the detector beside it.  Measured in round 5 (profiles/r05_full_step_ab.txt): 323.6 -> 323.5
        # env-steps/s at 8 environments (YOLOv7-E6E), 184 -> 185 with GroundingDINO, 783 -> 672 at 64: no gain -- the 8-environment
        # step is bound by the host's launch rate and its read-backs, not by GPU occupancy -- so it is OFF by default; the
        # equivalence test (tests/test_full_step_gpu.py) keeps the path honest.
        if os.environ.get("VLFM_VLM_BESIDE") is not None:      # diagnostic override (A/B runs): 0 = never, n = up to n environments
            concurrent_vlm_max_envs = int(os.environ["VLFM_VLM_BESIDE"])
        self.vlm_stream = torch.cuda.Stream(self.device) if overlap and n_envs <= concurrent_vlm_max_envs else None
        self.last_cosines: Optional[torch.Tensor] = None
        self.last_frontier_values: Optional[np.ndarray] = None
        # frontier selection of ITMPolicyV2 (stick-to-last rule, itm_policy.py:76-152), one selector per environment
        self.selectors = None
        if select_frontiers:
            from .policy_step import FrontierSelector

            self.selectors = [FrontierSelector() for _ in range(n_envs)]
        self.last_goals: Optional[np.ndarray] = None
        # PointNav controller (vlfm_amd.pointnav.WrappedPointNavResNetPolicy built for n_envs): the action towards the
        # selected frontier, one batched forward (base_objectnav_policy.py:243-283)
        self.pointnav = pointnav
        self.prev_goals = np.zeros((n_envs, 2))
        self.last_actions = None
        self.timers: Dict[str, List] = {}

    def reset(self) -> None:
        self.values.reset()
        if self.obstacles is not None:
            self.obstacles.reset()
        self.t = 0
        if self.selectors is not None:
            from .policy_step import FrontierSelector

            self.selectors = [FrontierSelector() for _ in range(self.E)]
        if self.object_maps is not None:
            for om in self.object_maps:
                om.reset()
        self.prev_goals = np.zeros((self.E, 2))
        if self.pointnav is not None:
            self.pointnav.reset()

    # ------------------------------------------------------------------------------------------ scripted detector head
    def _sightings_at(self, t_ep: int) -> List:
        """[(env slot, phrase, confidence, (cx, cy, ax, ay), depth)] of episode step ``t_ep`` (memoised: the painter asks when
        the frame is rendered, the step asks again when it runs)."""
        if t_ep not in self._sight_cache:
            if len(self._sight_cache) > 4096:
                self._sight_cache.clear()
            self._sight_cache[t_ep] = [(e, *sg) for e, i in enumerate(self.env_ids)
                                       for sg in self.sightings.at(i, t_ep, self.targets[e])]
        return self._sight_cache[t_ep]

    def _paint_sightings(self, t_ep: int, frames: torch.Tensor) -> torch.Tensor:
        """The scripted objects of step ``t_ep`` into the rendered depth frames: an ellipse at the object's depth wherever it is
        nearer than the wall / floor behind it (so both maps and the object cloud see one consistent scene)."""
        sg = self._sightings_at(t_ep)
        self._schedule(t_ep)
        if not sg:
            return frames
        idx = torch.tensor([s_[0] for s_ in sg], device=frames.device)
        m = ellipse_masks([s_[3] for s_ in sg], self.H, self.W, frames.device)
        d = torch.tensor([s_[4] for s_ in sg], dtype=frames.dtype, device=frames.device)[:, None, None]
        cur = frames[idx]
        frames[idx] = torch.where(m, torch.minimum(cur, d), cur)   # (an environment has at most one sighting per step)
        return frames

    CANDIDATES_PER_SIGHTING = 24

    def _inject_candidates(self, pred: torch.Tensor, in_hw, t_ep: int) -> torch.Tensor:
        """Write the scripted head's candidates into the detector's raw prediction [E, N, 5 + classes] (xywh in network-input pixels,
        objectness, class scores): per sighting a cluster of CANDIDATES_PER_SIGHTING boxes -- the scripted box with the scripted
        confidence and jittered, slightly less confident copies that the NMS has to suppress -- in the first rows of its frame;
        every other row keeps the network's own output (random weights: nothing passes the objectness gate)."""
        from .vlm.coco_classes import COCO_CLASSES

        sg = self._sightings_at(t_ep)
        if not sg:
            return pred
        K = self.CANDIDATES_PER_SIGHTING
        # the inverse of scale_coords (yolov7 [ext], as yolov7.py:99 calls it: one gain + centring pads, although the frame was
        # resized anisotropically -- the reference's quirk is kept): a box given in frame pixels comes back as itself, rounded
        gain = min(in_hw[0] / self.H, in_hw[1] / self.W)
        padx, pady = (in_hw[1] - self.W * gain) / 2, (in_hw[0] - self.H * gain) / 2
        rows = np.zeros((len(sg), K, pred.shape[2]), np.float32)
        # jitter of the suppressed copies: at most +-2 network pixels, scaled down for small boxes so that every copy keeps an IoU
        # above the NMS threshold (0.45) with the scripted box -- a copy that drifted below it would survive as a second detection
        jit = np.random.Generator(np.random.PCG64(4242 + t_ep)).uniform(-1.0, 1.0, size=(len(sg), K, 4)).astype(np.float32)
        jit[:, 0] = 0.0
        used = {}
        dst_e, dst_r = [], []
        for n, (e, phrase, conf, (cx, cy, ax, ay), _) in enumerate(sg):
            cls = COCO_CLASSES.index(phrase)
            amp = min(2.0, 0.05 * 2 * min(ax, ay) * gain)          # 5 % of the smaller side: IoU of a jittered copy >= ~0.8
            rows[n, :, 0] = cx * gain + padx + amp * jit[n, :, 0]
            rows[n, :, 1] = cy * gain + pady + amp * jit[n, :, 1]
            rows[n, :, 2] = 2 * ax * gain + amp * jit[n, :, 2]
            rows[n, :, 3] = 2 * ay * gain + amp * jit[n, :, 3]
            rows[n, :, 4] = 1.0
            rows[n, :, 5 + cls] = conf * np.concatenate([[1.0], np.linspace(0.97, 0.75, K - 1)])   # conf = objectness x class score
            base = used.get(e, 0)
            used[e] = base + K
            dst_e += [e] * K
            dst_r += list(range(base, base + K))
        # (written in place: the prediction is the detector's own scratch output of this step, produced under inference_mode)
        pred[torch.tensor(dst_e, device=pred.device), torch.tensor(dst_r, device=pred.device)] = \
            torch.from_numpy(rows.reshape(-1, rows.shape[2])).to(pred.device, pred.dtype)
        return pred

    def _scripted_detections(self, t_ep: int):
        """What the scripted head reports for every environment at this step, as the detector clients' ``ObjectDetections``
        (normalised xyxy boxes, f32, like yolov7.py:99-110 / grounding_dino.py:60-66)."""
        from .vlm.detections import ObjectDetections

        per = [[] for _ in range(self.E)]
        for (e, phrase, conf, (cx, cy, ax, ay), _) in self._sightings_at(t_ep):
            per[e].append(([(cx - ax) / self.W, (cy - ay) / self.H, (cx + ax) / self.W, (cy + ay) / self.H], conf, phrase))
        none = (torch.zeros((0, 4), dtype=torch.float32), torch.zeros(0, dtype=torch.float32))    # most environments, most steps
        return [ObjectDetections(none[0], none[1], [], image_source=None, fmt="xyxy") if not rows else
                ObjectDetections(torch.tensor([r[0] for r in rows], dtype=torch.float32).reshape(-1, 4),
                                 torch.tensor([r[1] for r in rows], dtype=torch.float32), [r[2] for r in rows],
                                 image_source=None, fmt="xyxy") for rows in per]

    # ------------------------------------------------------------------------------------------ object maps
    def _update_object_maps(self, dets, rgb: torch.Tensor, depth: torch.Tensor, tf: np.ndarray) -> None:
        """BaseObjectNavPolicy._update_object_map for every environment (base_objectnav_policy.py:285-352): class + confidence
        filters, ONE MobileSAM call for all surviving boxes of the batch (the reference re-encodes the frame per box too),
        ObjectPointCloudMap.update_map per mask (csrc/object_cloud.hip: erosion, back-projection, DBSCAN), update_explored
        per environment and step."""
        jobs = []
        for e, det in enumerate(dets):
            det.filter_by_class(self.targets[e].split("|"))
            det.filter_by_conf(self.det_threshold)
            # the reference multiplies the f32 row by an int64 ndarray: NumPy promotes to f64 first (:312)
            jobs += [(e, det.boxes[i].detach().cpu().numpy().astype(np.float64) * np.array([self.W, self.H, self.W, self.H]))
                     for i in range(len(det.logits))]
        self.object_stats["detections"] += len(jobs)
        self.last_masks = None
        if jobs:
            envs = [j[0] for j in jobs]
            boxes = np.stack([j[1] for j in jobs])
            masks = None
            if self.sam is not None:
                # the number of surviving boxes changes from step to step; MIOpen / hipBLASLt look every NEW batch size up (5 ms per
                # convolution the first time): the segmenter runs on a few fixed batch sizes, the tail padded with repeats
                n_pad = sam_batch_bucket(len(envs))
                pe = envs + [envs[-1]] * (n_pad - len(envs))
                pb = np.concatenate([boxes, np.repeat(boxes[-1:], n_pad - len(envs), axis=0)], axis=0)
                masks = self.sam.segment_bboxes(rgb[pe], torch.from_numpy(pb).to(torch.float32)[:, None, :])[:len(envs), 0]
            if masks is None or self.scripted_masks:
                # without pretrained weights the segmenter's logits mean nothing: the mask handed on is the box's inscribed ellipse
                # (the MobileSAM forward above still ran and is timed); also the stand-in when no segmenter is attached
                masks = ellipse_masks(np.stack([(boxes[:, 0] + boxes[:, 2]) / 2, (boxes[:, 1] + boxes[:, 3]) / 2,
                                                (boxes[:, 2] - boxes[:, 0]) / 2, (boxes[:, 3] - boxes[:, 1]) / 2], axis=1),
                                      self.H, self.W, self.device)
            self.last_masks = (envs, masks)
            self.object_stats["masks"] += len(jobs)
            for j, e in enumerate(envs):
                om = self.object_maps[e]
                before = len(om.clouds.get(self.targets[e], ()))
                om.update_map(self.targets[e], depth[e], masks[j], tf[e], MIN_DEPTH, MAX_DEPTH, self.fx, self.fy)
                self.object_stats["cloud_updates"] += int(len(om.clouds.get(self.targets[e], ())) != before)
        for e, om in enumerate(self.object_maps):
            if om.clouds:
                om.update_explored(tf[e], MAX_DEPTH, self.fov)     # cone_fov = get_fov(fx, width) (:349)

    # ------------------------------------------------------------------------------------------ act
    def _episode_steps(self, t_ep: int) -> np.ndarray:
        """Steps since each environment's last episode start (the policy's ``_num_steps``): the harness step for everybody
        without a script, the script's per-environment episode clock with one."""
        if self.sightings is None:
            return np.full(self.E, t_ep, np.int64)
        return self._schedule(t_ep)[0]

    def _schedule(self, t_ep: int):
        """(steps into the episode [E], episode ends with this step [E]) of the script at harness step ``t_ep``, memoised (the
        painter computes it when a frame is pre-rendered, outside a timed region)."""
        hit = self._sched_cache.get(t_ep)
        if hit is None:
            if len(self._sched_cache) > 4096:
                self._sched_cache.clear()
            loc = [self.sightings.locate(i, t_ep) for i in self.env_ids]
            hit = self._sched_cache[t_ep] = (np.array([l[1] for l in loc], np.int64),
                                            np.array([l[1] == l[2] - 1 for l in loc], bool))
        return hit

    def _end_episodes(self, t_ep: int) -> None:
        """Environments whose scripted episode ends with this step (the robot "arrived": the reference's episode ends on STOP):
        their maps, object map, selector and controller state are reset for the next episode, which starts in place."""
        self.last_episode_end = np.zeros(self.E, bool)
        if self.sightings is None:
            return
        done = np.flatnonzero(self._schedule(t_ep)[1]).tolist()
        if not done:
            return
        from .policy_step import FrontierSelector

        self.last_episode_end[done] = True
        self.values.reset(done)
        if self.obstacles is not None:
            self.obstacles.reset(done)
        for e in done:
            if self.object_maps is not None:
                self.object_maps[e].reset()
            if self.selectors is not None:
                self.selectors[e] = FrontierSelector()
        self.prev_goals[done] = 0.0
        if self.pointnav is not None:
            self.pointnav.reset(done)
        self.object_stats["episodes_ended"] = self.object_stats.get("episodes_ended", 0) + len(done)

    def _decide(self, wps: np.ndarray, env_of: np.ndarray, vals, poses: np.ndarray, t_ep: int):
        """BaseObjectNavPolicy.act's three modes for every environment (base_objectnav_policy.py:126-135): 12 initialisation
        turns, then the object goal if the object map has the target, else the best frontier (itm_policy.py:64-152).  Returns
        (modes, goals [E,2] (nan = none), stop_no_frontier [E])."""
        goals = np.full((self.E, 2), np.nan)
        modes, halt = [], np.zeros(self.E, bool)
        vals = np.asarray(vals, np.float64).reshape(-1) if vals is not None else np.zeros(0)
        bounds = np.searchsorted(env_of, np.arange(self.E + 1))
        ep_steps = self._episode_steps(t_ep)
        for e in range(self.E):
            target, robot_xy = self.targets[e], poses[e, :2]
            obj = None
            if self.object_maps is not None and self.object_maps[e].has_object(target):
                obj = self.object_maps[e].get_best_object(target, robot_xy)        # every step, like the reference (:123)
            if ep_steps[e] < 12:     # _done_initializing flips after the 12th call (habitat_policies.py:150-153)
                modes.append("initialize")
            elif obj is None:
                modes.append("explore")
                lo, hi = bounds[e], bounds[e + 1]
                if hi <= lo:
                    halt[e] = True          # "No frontiers found during exploration, stopping." itm_policy.py:64-67
                    continue
                order = np.argsort(-vals[lo:hi])       # sort_waypoints' descending order (value_map.py:183-186)
                pts = wps[lo:hi]
                goals[e], _ = self.selectors[e].choose(pts[order], [float(v) for v in vals[lo:hi][order]], pts, robot_xy)
            else:
                modes.append("navigate")
                goals[e] = obj[:2]
        return modes, goals, halt

    def _navigate(self, depth: torch.Tensor, modes: List[str], goals: np.ndarray, halt: np.ndarray, poses: np.ndarray):
        """BaseObjectNavPolicy._pointnav for every environment with a goal (base_objectnav_policy.py:243-283): a goal that moved
        by more than 0.1 m resets the controller, (rho, theta) in the robot frame (geometry_utils.py:9-34), STOP within
        ``pointnav_stop_radius`` of an OBJECT goal; ONE batched controller forward.  Environments that do not consult the
        controller this step (initialising, stopping, no frontier) keep its recurrent state and previous action untouched, as in
        the single-environment policy.  Returns the [E] action ids (TURN_LEFT while initialising, STOP where issued)."""
        from .policy_step import ACTION_STOP, ACTION_TURN_LEFT

        E = self.E
        have = ~np.isnan(goals[:, 0])
        moved = np.zeros(E, bool)
        moved[have] = np.linalg.norm(goals[have] - self.prev_goals[have], axis=1) > 0.1
        self.prev_goals[have] = goals[have]
        d = np.where(have[:, None], goals - poses[:, :2], 0.0)
        c, s = np.cos(-poses[:, 2]), np.sin(-poses[:, 2])
        lx, ly = c * d[:, 0] - s * d[:, 1], s * d[:, 0] + c * d[:, 1]
        rho, theta = np.hypot(lx, ly), np.arctan2(ly, lx)
        navigate = np.array([m == "navigate" for m in modes], bool)
        stop = halt | (have & navigate & (rho < self.stop_radius))
        run = have & ~stop
        self.last_stops, self.last_resets = stop, moved & have
        self.last_rho_theta = np.where(have[:, None], np.stack([rho, theta], axis=1), np.nan)
        if self.pointnav is None:
            return None
        pn = self.pointnav
        if moved.any():
            pn.reset(np.flatnonzero(moved))                     # (before the stop check, like the reference)
        keep = torch.from_numpy(~run).to(self.device)
        h0, a0 = pn.pointnav_test_recurrent_hidden_states.clone(), pn.pointnav_prev_actions.clone()
        rt = torch.from_numpy(np.stack([rho, theta], axis=1).astype(np.float32))
        acts = pn.act_on_depth(depth, rt, torch.from_numpy(~moved))
        pn.pointnav_test_recurrent_hidden_states[keep] = h0[keep]
        pn.pointnav_prev_actions[keep] = a0[keep]
        if not pn.discrete:
            return acts
        override = np.full(E, -1, np.int64)
        override[[m == "initialize" for m in modes]] = ACTION_TURN_LEFT
        override[stop] = ACTION_STOP
        ov = torch.from_numpy(override).to(self.device)
        return torch.where(ov >= 0, ov, acts.reshape(E).to(torch.int64))

    def warm_up_segmenter(self, max_boxes: Optional[int] = None) -> None:
        """Run the segmenter once at every batch size it can meet (``sam_batch_bucket``), outside any timed region: the library
        kernels' per-shape lookups happen here instead of in the first step that sees a size."""
        if self.sam is None:
            return
        top = sam_batch_bucket(max_boxes if max_boxes is not None else max(1, self.E // 2))
        rgb = self.rgb_pool[0]
        # on the stream the segmenter will run on (step()): MIOpen keeps its handle -- and with it the per-shape lookups -- per stream
        # (a first call on a fresh stream cost 9 ms per convolution)
        stream = self.obj_stream if (self.obj_stream is not None and self.detector is not None and self.object_maps is not None) \
            else torch.cuda.current_stream(self.device)
        torch.cuda.synchronize(self.device)
        with torch.cuda.stream(stream):
            for b in [b for b in SAM_BATCH_BUCKETS if b <= top]:
                idx = [i % self.E for i in range(b)]
                box = torch.tensor([[[0.3 * self.W, 0.3 * self.H, 0.7 * self.W, 0.8 * self.H]]] * b)
                self.sam.segment_bboxes(rgb[idx], box)
        torch.cuda.synchronize(self.device)

    def prepare(self, n_steps: int) -> None:
        """Render the depth frames of the next ``n_steps`` steps now (rooms world), so that a timed region that follows
        finds its inputs resident in HBM, as the benchmark contract asks."""
        if self.rooms is not None:
            self.rooms.prepare(self.t % self.episode_len, n_steps)

    def current_depth(self, n: int) -> torch.Tensor:
        """The depth frames of the first ``n`` environments at the current step (diagnostics: bench.count_stored_cells)."""
        if self.rooms is not None:
            return self.rooms.frame(self.t % self.episode_len)[:n]
        return self.depth_pool[self.t % self.depth_pool.shape[0]][:n].to(self.device)

    def fast_forward(self, n_steps: int) -> None:
        """Advance every episode by ``n_steps`` MAP-ONLY steps (stub cosines instead of the BLIP-2 forward, no detector /
        segmenter / controller): brings explored area, obstacle planes and contour lengths to a mid-episode state cheaply
        before a measurement, instead of timing the empty world of an episode's first steps."""
        saved = (self.blip2, self.detector, self.sam, self.selectors, self.pointnav, self.object_maps)
        self.blip2 = self.detector = self.sam = self.selectors = self.pointnav = self.object_maps = None
        try:
            for _ in range(n_steps):
                self.step()
        finally:
            self.blip2, self.detector, self.sam, self.selectors, self.pointnav, self.object_maps = saved

    def frontier_stats(self):
        """(mean, max) number of frontiers per environment at the last step (what the obstacle pipeline is working on)."""
        if self.obstacles is None or not self.obstacles.frontiers_ready:
            return None
        n = self.obstacles._h_counts.numpy()[:, 0]
        return [round(float(n.mean()), 2), int(n.max())]

    def check(self) -> None:
        """Raise what the reference would have raised inside the steps since the last check: IndexError for an obstacle
        point off the map (obstacle_map.py:101; the policy turns it into STOP, base_objectnav_policy.py:157-162), RuntimeError
        for an exhausted scratch capacity (never a silent wrong map).  Frontier-pipeline overflows already raise on the
        per-step frontier read-back; this adds the sticky flags of the depth passes.  One small D2H copy + sync: called
        at every episode end by step() and by the benchmark after its timed region, not per step."""
        if self.obstacles is not None:
            self.obstacles.check_status()
        if self.blip2 is not None and hasattr(self.blip2, "check_numerics"):
            self.blip2.check_numerics()
        if self.sam is not None and hasattr(self.sam, "check_numerics"):
            self.sam.check_numerics()

    def _log_finished_episodes(self) -> None:
        """One JSON file per finished episode in the reference's log format (vlfm/utils/log_saver.py:9-22) when
        ZSOS_LOG_DIR is set: what the reference's eval loop writes through episode_stats_logger.log_episode_stats."""
        if "ZSOS_LOG_DIR" not in os.environ:
            return
        from .utils.log_saver import is_evaluated, log_episode

        n_fr = self.obstacles.frontiers_px() if self.obstacles is not None and self.obstacles.frontiers_ready else None
        best = None
        if self.last_frontier_values is not None and len(self.last_frontier_values):
            best = float(np.max(self.last_frontier_values))
        for e, env_id in enumerate(self.env_ids):
            episode_id = self.episodes_done * len(self.env_ids) + e
            scene = f"synthetic{env_id:04d}"
            if is_evaluated(episode_id, scene):
                continue
            log_episode(episode_id, scene, {
                "target_object": self.targets[e], "num_steps": int(self.episode_len),
                "final_pose": [float(v) for v in self.pose_table[(self.t - 1) % self.episode_len][e]],
                "num_frontiers": int(len(n_fr[e])) if n_fr is not None else 0,
                "best_frontier_value_last_step": best})

    def step(self) -> None:
        if self.t and self.t % self.episode_len == 0:
            self.check()
            self._log_finished_episodes()
            self.episodes_done += 1
            self.reset()
        k = self.t % self.depth_pool.shape[0]
        kr = self.t % self.rgb_pool.shape[0]
        if self.host_inputs:
            depth = self.depth_dev.copy_(self.depth_pool[k], non_blocking=True)
            rgb = self.rgb_dev.copy_(self.rgb_pool[kr], non_blocking=True)
        elif self.rooms is not None:
            depth, rgb = self.rooms.frame(self.t % self.episode_len), self.rgb_pool[kr]
        else:
            depth, rgb = self.depth_pool[k], self.rgb_pool[kr]
        poses, tf = self.pose_table[self.t % self.episode_len], self.tf_table[self.t % self.episode_len]
        main = torch.cuda.current_stream(self.device)
        side = self.map_stream if self.map_stream is not None else main
        # ---- mapping, part 1 (side stream): one depth pass feeds both maps, then the obstacle/frontier pipeline
        side.wait_stream(main)  # the previous step's value update consumed the column-max keys
        with torch.cuda.stream(side):
            if self.obstacles is not None:
                colmax = self.obstacles.ingest(depth, tf, MIN_DEPTH, MAX_DEPTH, self.fx, self.fy, want_colmax=True)
                self.obstacles.update_after_ingest(tf, MAX_DEPTH, self.fov)
            else:
                colmax = self.values.column_max(depth)
        t_ep = self.t % self.episode_len
        # ---- detector (main stream).  With the object maps switched on it goes FIRST: its read-back is the step's first host
        # synchronisation anyway, and what follows it -- MobileSAM on the surviving boxes and one ObjectPointCloudMap.update_map per
        # mask, each a few small kernels and two host read-backs -- then runs on its own stream WHILE the BLIP-2 forward occupies the
        # GPU (17 ms of mostly idle GPU per 128-environment step before).  No result depends on the order: BLIP-2 sees the frames only.
        detector_first = self.detector is not None and self.object_maps is not None and self.obj_stream is not None
        dets = None

        def detect():
            # YOLOv7 takes the frames alone; GroundingDINO is prompted (MP3D-style caption, habitat_policies.py:139-141)
            scripted = self.sightings is not None and (self.detector is not None or self.object_maps is not None)
            if self.detector is None:
                d = None
            elif self.detector_is_prompted:
                d = self.detector.predict_batch(rgb, [self.gdino_caption])
            elif scripted and self.scripted_through_nms and hasattr(self.detector, "in_hw"):
                # the scripted head speaks THROUGH the detector's own post-processing: its candidates (a cluster of jittered boxes
                # per sighting, the scripted confidence on the best one) are written into the network's raw prediction, and
                # non_max_suppression / scale_coords / the rounding and normalisation of yolov7.py:91-110 produce the detections
                d = self.detector.predict_batch(rgb, pred_hook=lambda pred, in_hw: self._inject_candidates(pred, in_hw, t_ep))
                want = [0] * self.E
                for sg in self._sightings_at(t_ep):
                    want[sg[0]] += 1
                self.object_stats["head_mismatch"] = self.object_stats.get("head_mismatch", 0) + sum(
                    int(det.num_detections != w) for det, w in zip(d, want))
                return d
            else:
                d = self.detector.predict_batch(rgb)
            if scripted:
                d = self._scripted_detections(t_ep)     # the scripted HEAD: the network above ran (and is timed), its random logits are not used
            return d

        def perceive():
            return (self.blip2.cosine_batch_graphed(rgb, self.prompts) if self.graph_blip2
                    else self.blip2.cosine_batch(rgb, self.prompts))

        cos = None
        vlm_beside = detector_first and self.vlm_stream is not None and self.blip2 is not None
        if vlm_beside:
            self.vlm_stream.wait_stream(main)          # (the frames of this step are complete on the main stream)
            with torch.cuda.stream(self.vlm_stream):
                cos = perceive()
        if detector_first:
            dets = detect()
        # ---- perception (main stream): one batched BLIP-2 ITC forward for all resident envs
        if cos is not None:
            pass
        elif self.blip2 is not None:
            cos = perceive()
        else:
            cos = torch.from_numpy(self.stub_rng.uniform(0.15, 0.45, size=self.E)).to(self.device)
        self.last_cosines = cos
        if not detector_first:
            dets = detect()
        self.last_detections = dets
        if self.object_maps is not None and dets is not None:
            if detector_first:
                # (the frames were complete when the detector's read-back returned; nothing else on the main stream is an input)
                with torch.cuda.stream(self.obj_stream):
                    self._update_object_maps(dets, rgb, depth, tf)
                main.wait_stream(self.obj_stream)   # the next step may not repaint the frames under the segmenter
            else:
                self._update_object_maps(dets, rgb, depth, tf)
        elif self.sam is not None:
            # (legacy leg without object maps: MobileSAM on one fixed box for every ``sam_every``-th environment-step)
            sel = [e for e in range(self.E) if (self.t + e) % self.sam_every == 0]
            if sel:
                box = torch.tensor([[[0.3 * self.W, 0.3 * self.H, 0.7 * self.W, 0.8 * self.H]]] * len(sel))
                self.last_masks = self.sam.segment_bboxes(rgb[sel], box)
        # ---- frontiers back to the host (the policy needs them); waits for the side stream only, so the host-side
        # prologue of the value update overlaps the GPU's BLIP-2 work
        if self.obstacles is not None and self.obstacles.frontiers_ready:
            with torch.cuda.stream(side):
                wps, env_of = self.obstacles.frontier_list()
        else:
            ang = np.linspace(0, 2 * np.pi, self.n_frontiers, endpoint=False)
            wps = (poses[:, None, :2] + 1.5 * np.stack([np.cos(ang[None, :] + poses[:, 2:3]),
                                                        np.sin(ang[None, :] + poses[:, 2:3])], axis=2)).reshape(-1, 2)
            env_of = np.repeat(np.arange(self.E), self.n_frontiers)
        # ---- mapping, part 2 (main stream): value-map fusion needs the cosines and the column maxima
        main.wait_stream(side)
        if vlm_beside:
            main.wait_stream(self.vlm_stream)          # the cosines; and the next step may not repaint the frames under the ViT
            cos.record_stream(main)
        self.values.update(cos.reshape(self.E, 1), None, tf, MIN_DEPTH, MAX_DEPTH, self.fov, colmax=colmax)
        # ---- frontier scoring (ITMPolicyV2._sort_frontiers_by_value, radius 0.5 m)
        self.last_frontier_values = None
        if len(wps):
            self.last_frontier_values = self.values.waypoint_values(wps, env_of, 0.5)  # D2H sync: the policy needs it
        if self.selectors is not None:
            modes, goals, halt = self._decide(wps, env_of, self.last_frontier_values, poses, t_ep)
            self.last_modes, self.last_goals = modes, goals
            self.last_actions = self._navigate(depth, modes, goals, halt, poses)
            self.object_stats["env_steps"] += self.E
            for m in modes:
                self.object_stats["modes"][m] += 1
        self._end_episodes(t_ep)
        self.t += 1

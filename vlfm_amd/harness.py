"""Batched-episode harness: E independent synthetic ObjectNav episodes resident on ONE GPU, stepped together.

Replaces the reference's single-env Habitat eval loop (vlfm/utils/vlfm_trainer.py:164-174, which raises when
distributed, :65-66) for throughput measurement.  One `step()` performs, for every resident environment, the
per-step hot path of ITMPolicyV2 (vlfm/policy/itm_policy.py:251-261):

    _cache_observations -> ObstacleMap.update_map      (habitat_policies.py:193-201)      [depth ingest + obstacle kernels]
    _update_value_map   -> BLIP2ITMClient.cosine        (itm_policy.py:191-203)            [batched in-process BLIP-2 ITC]
                        -> ValueMap.update_map          (itm_policy.py:204-206)            [one launch for all envs]
    _explore            -> ValueMap.sort_waypoints      (itm_policy.py:263-267)            [disc medians per frontier]

Episodes are independent, so multi-GPU scaling is pure sharding (env e -> rank e mod world); the only collective is the
metric all-reduce in bench.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .mapping.base_map import require_gpu
from .mapping.value_map import ValueMapBatch, _bytes_to_device, _stream_ptr
from .synthetic import CAMERA_HEIGHT, MAX_DEPTH, MIN_DEPTH, Trajectory, camera_intrinsics, depth_frame, pose_to_tf, \
    rgb_frame

PROMPT = "Seems like there is a target_object ahead."  # vlfm/policy/base_objectnav_policy.py:377
TARGETS = ["chair", "bed", "potted plant", "toilet", "tv", "couch"]  # HM3D ObjectNav categories


class BatchedEpisodes:
    def __init__(self, n_envs: int, device=None, height: int = 480, width: int = 640, env_offset: int = 0,
                 blip2=None, use_blip2: bool = True, frame_pool: int = 4, map_size: int = 1000,
                 n_frontiers: int = 8, sync_explored: bool = False, obstacle: bool = True,
                 episode_len: int = 500) -> None:
        self.device = require_gpu(device)
        self.E, self.H, self.W, self.S = n_envs, height, width, map_size
        self.fx, self.fy, self.fov = camera_intrinsics(width)
        self.episode_len = episode_len
        self.env_ids = [env_offset + e for e in range(n_envs)]
        self.traj = [Trajectory(i) for i in self.env_ids]
        self.targets = [TARGETS[i % len(TARGETS)] for i in self.env_ids]
        self.prompts = [PROMPT.replace("target_object", t.replace("|", "/")) for t in self.targets]
        self.values = ValueMapBatch(n_envs, 1, map_size, use_max_confidence=False, device=self.device)
        self.n_frontiers = n_frontiers
        self.t = 0
        # synthetic observations live in HBM before the timed region starts (bench contract): a small pool of
        # distinct frames per env, cycled
        rng = np.random.Generator(np.random.PCG64(99991 + env_offset))
        self.depth_pool = torch.from_numpy(np.stack([
            np.stack([depth_frame(rng, height, width) for _ in range(n_envs)]) for _ in range(frame_pool)])).to(self.device)
        self.rgb_pool = torch.from_numpy(np.stack([
            np.stack([rgb_frame(rng, height, width) for _ in range(n_envs)]) for _ in range(frame_pool)])).to(self.device)
        self.blip2 = blip2
        if use_blip2 and blip2 is None:
            from .vlm.blip2itm import BLIP2ITM

            self.blip2 = BLIP2ITM(device=self.device)
        self.stub_rng = np.random.Generator(np.random.PCG64(7 + env_offset))
        self.obstacles = None
        if obstacle:
            from .mapping.obstacle_map import ObstacleMapBatch

            self.obstacles = ObstacleMapBatch(n_envs, min_height=0.61, max_height=0.88, agent_radius=0.18,
                                              area_thresh=1.5, size=map_size, device=self.device)
            if sync_explored:
                self.values.explored = self.obstacles.explored
        self.last_cosines: Optional[torch.Tensor] = None
        self.last_frontier_values: Optional[np.ndarray] = None
        self.timers: Dict[str, List] = {}

    def reset(self) -> None:
        self.values.reset()
        if self.obstacles is not None:
            self.obstacles.reset()
        self.traj = [Trajectory(i) for i in self.env_ids]
        self.t = 0

    def _timed(self, name: str):
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.timers.setdefault(name, []).append((start, stop))
        return start, stop

    def step(self, time_kernels: bool = False) -> None:
        if self.t and self.t % self.episode_len == 0:
            self.reset()
        k = self.t % self.depth_pool.shape[0]
        depth, rgb = self.depth_pool[k], self.rgb_pool[k]
        poses = [tr.step() for tr in self.traj]
        tf = np.stack([pose_to_tf(x, y, yaw) for (x, y, yaw) in poses])
        # ---- perception: one batched BLIP-2 ITC forward for all resident envs
        if self.blip2 is not None:
            cos = self.blip2.cosine_batch(rgb, self.prompts)
        else:
            cos = torch.from_numpy(self.stub_rng.uniform(0.15, 0.45, size=self.E)).to(self.device)
        self.last_cosines = cos
        # ---- mapping: one depth pass feeds both maps
        if time_kernels:
            a, b = self._timed("depth_ingest")
            a.record()
        if self.obstacles is not None:
            colmax = self.obstacles.ingest(depth, tf, MIN_DEPTH, MAX_DEPTH, self.fx, self.fy, want_colmax=True)
        else:
            colmax = self.values.column_max(depth)
        if time_kernels:
            b.record()
        if self.obstacles is not None:
            self.obstacles.update_after_ingest(tf, MAX_DEPTH, self.fov)
        if time_kernels:
            a, b = self._timed("value_map_update")
            a.record()
        self.values.update(cos.reshape(self.E, 1), None, tf, MIN_DEPTH, MAX_DEPTH, self.fov, colmax=colmax)
        if time_kernels:
            b.record()
        # ---- frontier scoring (ITMPolicyV2._sort_frontiers_by_value, radius 0.5 m)
        if self.obstacles is not None and self.obstacles.frontiers_ready:
            wps, env_of = self.obstacles.frontier_list()
        else:
            ang = np.linspace(0, 2 * np.pi, self.n_frontiers, endpoint=False)
            wps = np.concatenate([np.stack([x + 1.5 * np.cos(ang + yaw), y + 1.5 * np.sin(ang + yaw)], axis=1)
                                  for (x, y, yaw) in poses])
            env_of = np.repeat(np.arange(self.E), self.n_frontiers)
        if len(wps):
            self.last_frontier_values = self.values.waypoint_values(wps, env_of, 0.5)  # D2H sync: the policy needs it
        self.t += 1

    def kernel_ms(self) -> Dict[str, float]:
        torch.cuda.synchronize(self.device)
        return {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in self.timers.items() if v}

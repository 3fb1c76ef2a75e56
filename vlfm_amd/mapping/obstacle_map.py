"""vlfm.mapping.obstacle_map.ObstacleMap, MI355X-native (reference: /root/reference/vlfm/mapping/obstacle_map.py).

* :class:`ObstacleMapBatch` -- ``n_envs`` obstacle / navigable / explored planes resident in HBM, bit-packed, updated by
  one kernel pipeline per step for all environments (csrc/depth_ingest.hip + csrc/obstacle_map.hip).
* :class:`ObstacleMap` -- the reference's class signature (obstacle_map.py:25-34, :55-66) as a drop-in for
  ``BaseObjectNavPolicy`` (base_objectnav_policy.py:86-92) and ``HabitatMixin._cache_observations``
  (habitat_policies.py:193-203).

Scalar prologues (agent cell, cone angles) are evaluated in NumPy exactly as the reference does; every per-pixel step is
a HIP kernel.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Any, List, Optional, Sequence, Union

import numpy as np

from .. import _lib
from .base_map import BaseMap, require_gpu
from .value_map import INGEST_DTYPE, UploadRing, _stream_ptr, wait_stream


def _wrap_heading(theta):
    return (theta + np.pi) % (2 * np.pi) - np.pi


class ObstacleMapBatch:
    CAP_PTS = 16384       # border points per environment and scan
    CAP_CONTOURS = 2048
    CAP_FRONTIERS = 256
    READ_FRONTIERS = 32   # frontiers per environment fetched by the fast read-back path (more -> one full copy)
    HOLE_CAP_PTS = 1 << 17      # border points per depth image in fill_small_holes
    HOLE_CAP_CONTOURS = 1 << 15

    def __init__(self, n_envs: int, min_height: float, max_height: float, agent_radius: float, area_thresh: float = 3.0,
                 hole_area_thresh: int = 100000, size: int = 1000, pixels_per_meter: int = 20, device=None) -> None:
        import torch

        self.device = require_gpu(device)
        L = _lib.lib()
        self.n_envs, self.size, self.pixels_per_meter = n_envs, size, pixels_per_meter
        self.stride = (size + 31) // 32
        self._min_height, self._max_height = min_height, max_height
        self._area_thresh_in_pixels = area_thresh * (pixels_per_meter ** 2)
        self._hole_area_thresh = hole_area_thresh
        k = pixels_per_meter * agent_radius * 2
        self.kernel_size = int(k) + (int(k) % 2 == 0)  # obstacle_map.py:43-45
        z = lambda: torch.zeros((n_envs, size, self.stride), dtype=torch.int32, device=self.device)  # noqa: E731
        self.obstacle_bits, self.navigable_bits, self.explored_bits = z(), z(), z()
        self.bbox = torch.tensor([[size, -1, size, -1]] * n_envs, dtype=torch.int32, device=self.device)
        nbytes = L.vlfm_obstacle_scratch_bytes(n_envs, size, self.CAP_PTS, self.CAP_CONTOURS)
        self.scratch = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.frontiers_px_dev = torch.zeros((n_envs, self.CAP_FRONTIERS, 2), dtype=torch.float64, device=self.device)
        self.counts = torch.zeros((n_envs, 4), dtype=torch.int32, device=self.device)
        self.colmax_keys = None
        self.status = torch.zeros((n_envs, 2), dtype=torch.int32, device=self.device)
        self._ring_ingest = UploadRing(self.device, n_envs * INGEST_DTYPE.itemsize, slots=8)
        self._ring_fog = UploadRing(self.device, n_envs * ctypes.sizeof(_lib.FogParams), slots=8)
        self.frontiers_ready = False
        self._explored_u8 = None
        self._journal_dirty = False   # the scatter journal holds entries no fill_small_holes launch has consumed
        # Dirty windows (csrc/obstacle_map.hip: navigable_kernel / frontier_prepare_kernel), per slot, inclusive
        # (y0, y1, x0, x1), empty when y1 < y0.  `_dirty_obst`: cells whose obstacle bit may have changed since `navigable`
        # was last recomputed (union of the reach windows of the frames ingested); `_dirty_nav`: cells whose `navigable` bit
        # may have changed since the last explore step.  A fresh slot starts with the whole map (first pass = the
        # reference's full-map pass).  VLFM_FULL_PLANES=1 hands over NULL windows: the full-plane kernels, for A/B runs.
        self._dirty_obst = np.tile(np.array([0, size - 1, 0, size - 1], np.int32), (n_envs, 1))
        self._dirty_nav = self._dirty_obst.copy()
        # host mirror of the device's bounding box of everything ever revealed (fog_of_war_kernel: agent cell +- (R + 2))
        self._bbox_host = np.tile(np.array([0, -1, 0, -1], np.int32), (n_envs, 1))
        self._ring_win = UploadRing(self.device, n_envs * 48, slots=8)
        import os

        self.full_planes = os.environ.get("VLFM_FULL_PLANES", "0") == "1"
        # read-back path of the step: fixed-size staging + pinned host buffers, so a step never allocates and never
        # hands the runtime a pageable destination (which it would have to pin on the fly)
        self._d_fr_stage = torch.zeros((n_envs, self.READ_FRONTIERS, 2), dtype=torch.float64, device=self.device)
        self._h_fr = torch.zeros((n_envs, self.READ_FRONTIERS, 2), dtype=torch.float64).pin_memory()
        self._h_counts = torch.zeros((n_envs, 4), dtype=torch.int32).pin_memory()
        self._h_status = torch.zeros((n_envs, 2), dtype=torch.int32).pin_memory()

    # ------------------------------------------------------------------------------------------ state
    def reset(self, env_ids: Optional[Sequence[int]] = None) -> None:
        import torch

        idx = list(range(self.n_envs)) if env_ids is None else list(env_ids)
        for t in (self.obstacle_bits, self.navigable_bits, self.explored_bits):
            t[idx] = 0
        self.bbox[idx] = torch.tensor([self.size, -1, self.size, -1], dtype=torch.int32, device=self.device)
        self.counts[idx] = 0
        self.frontiers_ready = False
        self._explored_u8 = None
        self._dirty_obst[idx] = (0, self.size - 1, 0, self.size - 1)   # zeroed planes: everything is to be recomputed
        self._dirty_nav[idx] = (0, self.size - 1, 0, self.size - 1)
        self._bbox_host[idx] = (0, -1, 0, -1)

    @staticmethod
    def _union(into: np.ndarray, idx: np.ndarray, win: np.ndarray) -> None:
        """into[idx] |= win for inclusive (y0, y1, x0, x1) windows (empty: y1 < y0); idx holds distinct slots."""
        cur = into[idx]
        empty_cur, empty_new = cur[:, 1] < cur[:, 0], win[:, 1] < win[:, 0]
        out = np.stack([np.minimum(cur[:, 0], win[:, 0]), np.maximum(cur[:, 1], win[:, 1]),
                        np.minimum(cur[:, 2], win[:, 2]), np.maximum(cur[:, 3], win[:, 3])], axis=1)
        out[empty_cur] = win[empty_cur]
        out[empty_new] = cur[empty_new]
        into[idx] = out

    def _note_ingest(self, env: np.ndarray, tf: np.ndarray, reach_px: float) -> None:
        """Every obstacle bit a frame can touch lies within ``reach_px`` cells of its camera cell (texel distance from the
        camera <= max_depth * sqrt(1 + (W/2fx)^2 + (H/2fy)^2), rigid transform, one cell of rounding).  A window that
        leaves the map is replaced by the whole map: out-of-range rows / columns wrap like NumPy's negative indices
        (obstacle_map.py:101), and so is a camera transform whose rotation block is not orthonormal."""
        S = self.size
        R = tf[:, :3, :3]
        rigid = np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).reshape(len(tf), -1).max(axis=1) < 1e-9
        cell = np.rint(tf[:, :2, 3] * self.pixels_per_meter) + S // 2        # (x -> row, y -> S - col), base_map.py:44-46
        row, col = cell[:, 0], S - cell[:, 1]
        r = int(np.ceil(reach_px)) + 2
        win = np.stack([row - r, row + r, col - r, col + r], axis=1)
        inside = rigid & np.isfinite(win).all(axis=1) & (win[:, 0] >= 0) & (win[:, 1] <= S - 1) & (win[:, 2] >= 0) & \
            (win[:, 3] <= S - 1)
        win = np.where(inside[:, None], win, np.array([0, S - 1, 0, S - 1], np.float64)[None]).astype(np.int32)
        self._union(self._dirty_obst, env, win)

    def _take_windows(self, env: np.ndarray, update_obstacles: bool, explore: bool, agent_px=None, fog_radius: int = 0,
                      skipped=None):
        """([n, 12] int32 windows, workgroups for navigable_kernel, workgroups for frontier_prepare_kernel) for
        vlfm_obstacle_map_update_batched, and the bookkeeping that goes with handing them over.  ``agent_px`` [n, 2] (x, y)
        = the agent cells of an explore call: the reveal touches agent +- (fog_radius + 2).  ``skipped`` [n] bool marks
        observations whose fog params carry n_poly <= 0: the kernels do nothing for them, so their pending refresh windows stay
        pending (include/vlfm_amd.h, d_windows)."""
        S, n = self.size, len(env)
        empty = np.array([0, -1, 0, -1], np.int32)
        out = np.tile(np.concatenate([empty, empty, empty]), (n, 1))
        if update_obstacles:
            g = self.kernel_size // 2
            w = self._dirty_obst[env].copy()
            live = w[:, 1] >= w[:, 0]
            w[:, 0] = np.maximum(w[:, 0] - g, 0); w[:, 1] = np.minimum(w[:, 1] + g, S - 1)
            w[:, 2] = np.maximum(w[:, 2] - g, 0); w[:, 3] = np.minimum(w[:, 3] + g, S - 1)
            w[~live] = empty
            out[:, 0:4] = w
            self._union(self._dirty_nav, env, w)
            self._dirty_obst[env] = empty
        if explore:
            out[:, 4:8] = self._bbox_host[env]                     # explored bits live inside the box of the steps so far
            r = int(fog_radius) + 2
            ax, ay = np.asarray(agent_px[:, 0], np.int64), np.asarray(agent_px[:, 1], np.int64)
            fog = np.stack([np.maximum(ay - r, 0), np.minimum(ay + r, S - 1), np.maximum(ax - r, 0),
                            np.minimum(ax + r, S - 1)], axis=1).astype(np.int32)       # (window side 2 r + 1 = 2 R + 5)
            self._union(self._bbox_host, env, fog)
            box = self._bbox_host[env].copy()
            live = box[:, 1] >= box[:, 0]
            box[:, 0] = np.maximum(box[:, 0] - 3, 0); box[:, 1] = np.minimum(box[:, 1] + 3, S - 1)
            box[:, 2] = np.maximum(box[:, 2] - 3, 0); box[:, 3] = np.minimum(box[:, 3] + 3, S - 1)
            box[~live] = empty
            refresh = self._dirty_nav[env].copy()
            tmp = np.tile(empty, (n, 1))
            tmp[:] = refresh
            self._union(tmp, np.arange(n), box)
            out[:, 8:12] = tmp
            done = env if skipped is None else env[~np.asarray(skipped, bool)]
            self._dirty_nav[done] = empty

        def blocks(*wins) -> int:
            """256-word workgroups for the largest bounding union of the given windows over the batch."""
            u = np.tile(empty, (n, 1))
            for w_ in wins:
                self._union(u, np.arange(n), w_)
            rows = np.maximum(u[:, 1] - u[:, 0] + 1, 0)
            words = np.maximum((u[:, 3] >> 5) - (u[:, 2] >> 5) + 1, 0)
            return int(max(1, -(-int((rows * words).max()) // 256)))

        return out.astype(np.int32), blocks(out[:, 0:4], out[:, 4:8]), blocks(out[:, 8:12])

    def _unpack(self, bits):
        import torch

        out = torch.empty((self.n_envs, self.size, self.size), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().vlfm_bits_unpack(bits.data_ptr(), out.data_ptr(), self.n_envs, self.size, self.size,
                                                  _stream_ptr()), "bits_unpack")
        return out

    @property
    def explored(self):
        """[n_envs,S,S] uint8 view of ObstacleMap.explored_area (unpacked on demand, cached per step)."""
        if self._explored_u8 is None:
            self._explored_u8 = self._unpack(self.explored_bits)
        return self._explored_u8

    # ------------------------------------------------------------------------------------------ step, part 1
    def ingest(self, depth, tf, min_depth: float, max_depth: float, fx: float, fy: float, want_colmax: bool = False,
               env_ids: Optional[Sequence[int]] = None, update_obstacles: bool = True):
        """obstacle_map.py:86-101 for n observations: one pass over the depth images (shared with the value map's
        column maximum when ``want_colmax``).  Returns the column-max key buffer (or None)."""
        import torch

        n, H, W = depth.shape
        tf = np.asarray(tf, np.float64).reshape(n, 4, 4)
        assert np.array_equal(tf[:, 3, :], np.tile([0.0, 0, 0, 1], (n, 1))), "camera transforms must be affine"
        prm = np.zeros(n, INGEST_DTYPE)
        prm["tf"] = tf[:, :3, :].reshape(n, 12)
        prm["depth_scale"], prm["depth_offset"], prm["depth_max"] = max_depth - min_depth, min_depth, max_depth
        prm["fx"], prm["fy"] = fx, fy
        prm["min_height"], prm["max_height"] = self._min_height, self._max_height
        prm["env"] = np.arange(n) if env_ids is None else np.asarray(env_ids)
        assert int(prm["env"].max()) < self.n_envs and int(prm["env"].min()) >= 0, "environment slot out of range"
        prm["scatter"] = (1 if update_obstacles else 0) | (2 if self._hole_area_thresh == -1 else 0)
        if update_obstacles:
            reach_px = max_depth * float(np.sqrt(1.0 + (W / 2 / fx) ** 2 + (H / 2 / fy) ** 2)) * self.pixels_per_meter
            self._note_ingest(np.asarray(prm["env"], np.int64), tf, reach_px)
        keys = None
        if want_colmax:
            if self.colmax_keys is None or self.colmax_keys.shape != (max(n, self.n_envs), W):
                self.colmax_keys = torch.zeros((max(n, self.n_envs), W), dtype=torch.int32, device=self.device)
            keys = self.colmax_keys
        L = _lib.lib()
        fill = update_obstacles and self._hole_area_thresh != -1
        if fill:
            prm["scatter"] |= 4  # zero texels wait for fill_small_holes
            # the speculative pass journals the bits it sets per observation; two observations of one slot in the same
            # launch would hide each other's first-time bits (the reference's multi-camera loop is sequential anyway)
            assert len(np.unique(prm["env"])) == n, "one observation per environment slot and call"
        with torch.cuda.device(self.device):
            d_prm = self._ring_ingest.upload(prm)
            if not fill:
                # hole_area_thresh == -1 (every zero texel -> 1.0, obstacle_map.py:87-89) or no obstacle update:
                # one pass over the depth images
                _lib.check(L.vlfm_depth_ingest_batched(depth.data_ptr(), n, H, W, d_prm.data_ptr(),
                                                       keys.data_ptr() if keys is not None else None,
                                                       self.obstacle_bits.data_ptr() if update_obstacles else None,
                                                       self.size, self.pixels_per_meter, self.status.data_ptr(),
                                                       None, None, None, _stream_ptr()), "depth_ingest")
            else:
                # fill_small_holes (img_utils.py:361-390) sits between reading the depth and scattering it, but it only
                # decides the fate of texels inside zero regions: the single streaming pass SPECULATIVELY places every
                # non-zero texel (journalling the obstacle bits it is the first to set), reduces the column maxima and
                # emits the (depth == 0) bit plane + "has zeros" flag; the hole kernel exits immediately for images
                # without zeros; the zeros that survive it (large holes keep depth 0 -> z = min_depth) are then placed
                # from the two bit planes.  Only an "island" frame -- valid texels enclosed by a small hole, which the
                # reference's filled contour rewrites to 1.0 (img_utils.py:385-388) -- has its journalled bits taken
                # back and its valid texels outside the filled area placed again; every other image is read exactly once.
                # reach of a texel in the map plane: |(z, x, y)| <= max_depth * sqrt(1 + (W/2fx)^2 + (H/2fy)^2)
                reach = max_depth * float(np.sqrt(1.0 + (W / 2 / fx) ** 2 + (H / 2 / fy) ** 2)) * self.pixels_per_meter
                cap = int(min(self.size * self.size, (2 * int(np.ceil(reach)) + 3) ** 2))
                holes, filled, scratch, counts, journal = self._hole_buffers(n, H, W, cap)
                jref = ctypes.byref(journal)
                if self._journal_dirty:
                    # an earlier step launched the speculative pass but never reached fill_small_holes (which consumes the
                    # journal and resets its counters): entries of that step must not be undone by a later island frame
                    self._journal_count.zero_()
                self._journal_dirty = True
                _lib.check(L.vlfm_depth_ingest_batched(depth.data_ptr(), n, H, W, d_prm.data_ptr(),
                                                       keys.data_ptr() if keys is not None else None,
                                                       self.obstacle_bits.data_ptr(), self.size,
                                                       self.pixels_per_meter, self.status.data_ptr(),
                                                       holes.data_ptr(), None, jref, _stream_ptr()), "depth_ingest")
                _lib.check(L.vlfm_fill_small_holes_batched(holes.data_ptr(), self.status.data_ptr(), n, H, W,
                                                           float(self._hole_area_thresh), scratch.data_ptr(),
                                                           scratch.numel(), self.HOLE_CAP_PTS, self.HOLE_CAP_CONTOURS,
                                                           filled.data_ptr(), counts.data_ptr(), d_prm.data_ptr(),
                                                           self.obstacle_bits.data_ptr(), self.size, jref,
                                                           _stream_ptr()), "fill_small_holes")
                self._journal_dirty = False   # consumed: fill_small_holes_kernel leaves every counter at zero
                _lib.check(L.vlfm_depth_scatter_holes_batched(d_prm.data_ptr(), n, H, W, holes.data_ptr(),
                                                              filled.data_ptr(), counts.data_ptr(),
                                                              self.obstacle_bits.data_ptr(), self.size,
                                                              self.pixels_per_meter, self.status.data_ptr(),
                                                              depth.data_ptr(), _stream_ptr()), "depth_scatter_holes")
        return keys

    def _hole_buffers(self, n: int, H: int, W: int, journal_cap: int):
        import torch

        key = (max(n, self.n_envs), H, W, journal_cap)
        if getattr(self, "_hole_key", None) != key:
            m, hw = key[0], (W + 31) // 32
            self._hole_bits = torch.zeros((m, H, hw), dtype=torch.int32, device=self.device)
            self._filled_bits = torch.zeros((m, H, hw), dtype=torch.int32, device=self.device)
            nbytes = _lib.lib().vlfm_hole_scratch_bytes(m, H, W, self.HOLE_CAP_PTS, self.HOLE_CAP_CONTOURS)
            self._hole_scratch = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
            self._hole_counts = torch.zeros((m, 4), dtype=torch.int32, device=self.device)
            self._journal_cells = torch.empty((m, journal_cap), dtype=torch.int32, device=self.device)
            self._journal_count = torch.zeros(m, dtype=torch.int32, device=self.device)
            self._journal = _lib.ScatterJournal(self._journal_cells.data_ptr(), self._journal_count.data_ptr(),
                                                journal_cap, 0)
            self._hole_key = key
        return self._hole_bits, self._filled_bits, self._hole_scratch, self._hole_counts, self._journal

    # ------------------------------------------------------------------------------------------ step, part 2
    def fog_params(self, tf, max_depth: float, topdown_fov: float, env_ids=None, explore=None):
        """Scalar prologue of obstacle_map.py:115-124 + reveal_fog_of_war's first lines, in NumPy like the reference."""
        tf = np.asarray(tf, np.float64).reshape(-1, 4, 4)
        n = tf.shape[0]
        # BaseMap._xy_to_px (rint) of the camera position -> (col, row)
        px = np.rint(tf[:, :2, 3][:, ::-1] * self.pixels_per_meter) + self.size // 2
        px[:, 0] = self.size - px[:, 0]
        agent = np.ascontiguousarray(px.astype(np.int32))
        yaw = np.arctan2(tf[:, 1, 0], tf[:, 0, 0])
        current_angle = -yaw
        angle_cv2 = np.rad2deg(_wrap_heading(-current_angle + np.pi / 2))
        rot = np.ascontiguousarray(np.stack([np.cos(-angle_cv2), np.sin(-angle_cv2)], axis=1))  # degrees-as-radians [ext]
        out = (_lib.FogParams * n)()
        env = None if env_ids is None else np.ascontiguousarray(np.asarray(env_ids, np.int32))
        ex = None if explore is None else np.ascontiguousarray(np.asarray(explore, np.int32))
        angle_cv2 = np.ascontiguousarray(angle_cv2)
        _lib.check(_lib.lib().vlfm_fog_params_host(agent.ctypes.data, angle_cv2.ctypes.data, rot.ctypes.data,
                                                  float(np.rad2deg(topdown_fov)),
                                                  float(max_depth * self.pixels_per_meter),
                                                  env.ctypes.data if env is not None else None,
                                                  ex.ctypes.data if ex is not None else None, n,
                                                  ctypes.addressof(out)), "fog_params_host")
        return out

    def update_after_ingest(self, tf, max_depth: float, topdown_fov: float, env_ids=None, explore: bool = True,
                            update_obstacles: bool = True) -> None:
        import torch

        prm = self.fog_params(tf, max_depth, topdown_fov, env_ids)
        n = len(prm)
        if env_ids is not None:  # the explore pipeline owns a slot's planes for the whole launch
            assert len(set(int(e) for e in env_ids)) == n, "one observation per environment slot and call"
        env = np.arange(n) if env_ids is None else np.asarray(env_ids, np.int64)
        agent = np.array([[q.ax, q.ay] for q in prm], np.int64) if explore else None
        windows, nb_nav, nb_prep = self._take_windows(env, update_obstacles, explore, agent,
                                                      int(max_depth * self.pixels_per_meter),
                                                      skipped=np.array([q.n_poly <= 0 for q in prm], bool))
        with torch.cuda.device(self.device):
            d_prm = self._ring_fog.upload(prm)
            d_win = None if self.full_planes else self._ring_win.upload(windows)
            _lib.check(_lib.lib().vlfm_obstacle_map_update_batched(
                d_prm.data_ptr(), n, self.obstacle_bits.data_ptr(), self.navigable_bits.data_ptr(),
                self.explored_bits.data_ptr(), self.bbox.data_ptr(), self.n_envs, self.size, self.kernel_size,
                int(max_depth * self.pixels_per_meter), float(self._area_thresh_in_pixels), self.scratch.data_ptr(),
                self.scratch.numel(), self.CAP_PTS, self.CAP_CONTOURS, self.frontiers_px_dev.data_ptr(),
                self.CAP_FRONTIERS, self.counts.data_ptr(), int(update_obstacles), int(explore),
                d_win.data_ptr() if d_win is not None else None, nb_nav, nb_prep, _stream_ptr()), "obstacle_map_update")
        self.frontiers_ready = bool(explore)
        self._explored_u8 = None

    # ------------------------------------------------------------------------------------------ read-back
    def frontiers_px(self) -> List[np.ndarray]:
        """Per environment: (F,2) f64 pixel coordinates (x,y) == ObstacleMap._frontiers_px.  One D2H copy."""
        counts, fr = self._read_frontiers()
        return [fr[e, :counts[e, 0]].copy() for e in range(self.n_envs)]

    def _read_frontiers(self):
        import torch

        with torch.cuda.device(self.device):
            self._d_fr_stage.copy_(self.frontiers_px_dev[:, :self.READ_FRONTIERS])
            self._h_counts.copy_(self.counts, non_blocking=True)
            self._h_fr.copy_(self._d_fr_stage, non_blocking=True)
            wait_stream()
        counts = self._h_counts.numpy()
        if (counts[:, 1] != 0).any():
            raise RuntimeError("obstacle-map scratch capacity exceeded (CAP_PTS/CAP_CONTOURS/CAP_FRONTIERS)")
        if len(counts) and int(counts[:, 0].max()) > self.READ_FRONTIERS:
            return counts, self.frontiers_px_dev.cpu().numpy()
        return counts, self._h_fr.numpy()

    def px_to_xy(self, px: np.ndarray) -> np.ndarray:
        q = px.copy()
        q[:, 0] = self.size - q[:, 0]
        pts = (q - self.size // 2) / self.pixels_per_meter
        return pts[:, ::-1]

    def frontier_list(self):
        """All frontiers of all environments as (xy [M,2], env index [M]) for ValueMapBatch.waypoint_values."""
        counts, fr = self._read_frontiers()
        keep = np.arange(fr.shape[1])[None, :] < counts[:, :1]          # [n_envs, top]
        env_of = np.nonzero(keep)[0]
        px = fr[keep]                                                    # env-major, frontier order preserved
        return (self.px_to_xy(px) if len(px) else np.zeros((0, 2))), env_of

    def check_status(self) -> None:
        import torch

        with torch.cuda.device(self.device):
            self._h_status.copy_(self.status, non_blocking=True)
            wait_stream()
        st = self._h_status.numpy().copy()
        self.status.zero_()
        if (st[:, 0] != 0).any():
            raise IndexError("index out of bounds: obstacle point fell off the map (obstacle_map.py:101)")
        if getattr(self, "_hole_key", None) is not None:
            hc = self._hole_counts.cpu().numpy()
            if (hc[:, 2] != 0).any():
                raise RuntimeError("fill_small_holes scratch capacity exceeded (HOLE_CAP_PTS/HOLE_CAP_CONTOURS or the "
                                   "scatter journal)")


class ObstacleMap(BaseMap):
    """Drop-in for vlfm.mapping.obstacle_map.ObstacleMap."""

    radius_padding_color: tuple = (100, 100, 100)

    def __init__(self, min_height: float, max_height: float, agent_radius: float, area_thresh: float = 3.0,
                 hole_area_thresh: int = 100000, size: int = 1000, pixels_per_meter: int = 20, device=None):
        super().__init__(size, pixels_per_meter)
        self._batch = ObstacleMapBatch(1, min_height, max_height, agent_radius, area_thresh, hole_area_thresh, size,
                                       pixels_per_meter, device)
        self._min_height, self._max_height = min_height, max_height
        self._area_thresh_in_pixels = self._batch._area_thresh_in_pixels
        self._hole_area_thresh = hole_area_thresh
        self._navigable_kernel = np.ones((self._batch.kernel_size, self._batch.kernel_size), np.uint8)
        self._frontiers_px: np.ndarray = np.array([])
        self.frontiers: np.ndarray = np.array([])

    # host snapshots of the HBM planes for callers that read them
    @property
    def _map(self) -> np.ndarray:
        return self._batch._unpack(self._batch.obstacle_bits)[0].cpu().numpy().astype(bool)

    @property
    def _navigable_map(self) -> np.ndarray:
        return self._batch._unpack(self._batch.navigable_bits)[0].cpu().numpy().astype(np.int64)

    @property
    def explored_area(self) -> np.ndarray:
        return self._batch.explored[0].cpu().numpy().astype(bool)

    def explored_bits_device(self):
        """[1,S,ceil(S/32)] bit-packed explored area in HBM for ValueMap's explored-area synchronisation
        (value_map.py:369-375)."""
        return self._batch.explored_bits

    def reset(self) -> None:
        super().reset()
        self._batch.reset()
        self._frontiers_px = np.array([])
        self.frontiers = np.array([])

    def update_map(self, depth: Union[np.ndarray, Any], tf_camera_to_episodic: np.ndarray, min_depth: float,
                   max_depth: float, fx: float, fy: float, topdown_fov: float, explore: bool = True,
                   update_obstacles: bool = True) -> None:
        import torch

        tf = np.asarray(tf_camera_to_episodic, np.float64)
        if update_obstacles:
            if not torch.is_tensor(depth):
                depth = torch.from_numpy(np.ascontiguousarray(depth, np.float32)).to(self._batch.device)
            depth = depth.reshape(1, depth.shape[-2], depth.shape[-1]).contiguous()
            self._batch.ingest(depth, tf[None], min_depth, max_depth, fx, fy)
            self._batch.check_status()
        self._batch.update_after_ingest(tf[None], max_depth, topdown_fov, explore=explore,
                                        update_obstacles=update_obstacles)
        if not explore:
            return
        self._frontiers_px = self._batch.frontiers_px()[0]
        if len(self._frontiers_px) == 0:
            self._frontiers_px = np.array([])
            self.frontiers = np.array([])
        else:
            self.frontiers = self._px_to_xy(self._frontiers_px)

    def visualize(self) -> np.ndarray:
        vis = np.ones((self.size, self.size, 3), dtype=np.uint8) * 255
        vis[self.explored_area == 1] = (200, 255, 200)
        vis[self._navigable_map == 0] = self.radius_padding_color
        vis[self._map == 1] = (0, 0, 0)
        return vis[::-1].copy()

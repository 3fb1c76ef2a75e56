"""vlfm.mapping.value_map.ValueMap, MI355X-native (reference: /root/reference/vlfm/mapping/value_map.py).

Two layers:

* :class:`ValueMapBatch` -- ``n_envs`` value maps resident in HBM on one GPU, updated by ONE kernel launch per step
  for all environments (depth ingest -> profile polygon -> LDS coverage -> rotate/place/fuse).  This is what the
  batched-episode harness drives.
* :class:`ValueMap` -- the reference's class signature (value_map.py:44-51, :100-108, :146-148) as a drop-in for
  ``BaseITMPolicy`` (itm_policy.py:48-54, :191-211, :263-267); a thin single-slot view of a ValueMapBatch.

PyTorch is used for device memory and streams only; every per-pixel operation is a hand-written gfx950 kernel in
``vlfm_amd/csrc`` reached through the C ABI of ``include/vlfm_amd.h``.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .. import _lib
from .base_map import BaseMap, require_gpu


def _stream_ptr() -> int:
    import torch

    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())   # (see vlm/ops.py:_stream)


_WAIT_EVENTS: Dict[int, Any] = {}


def wait_stream() -> None:
    """Wait for the current stream on a reusable event.  Whether the wait spins or sleeps is the DEVICE's schedule flag
    (hipDeviceScheduleBlockingSync, set by _lib.host_wait_blocking / vlfm_host_wait_mode): measured on MI355X
    (tools/host_wait_probe.py) hipEventBlockingSync on the event alone still holds a core for the whole wait, the device
    flag brings the waiting thread to 0.00 cores.  The per-step read-backs sit behind ~100 ms of queued perception
    work, so a spinning rank costs a full host core (`host` block of bench.py)."""
    import torch

    dev = torch.cuda.current_device()
    ev = _WAIT_EVENTS.get(dev)
    if ev is None:
        ev = _WAIT_EVENTS[dev] = torch.cuda.Event(blocking=True)
    ev.record()
    ev.synchronize()


def _bytes_to_device(buf, device):
    """Upload a ctypes array / bytes object as a uint8 tensor (8-byte aligned by the caching allocator)."""
    import torch

    host = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8)
    return host.to(device, non_blocking=False)


class UploadRing:
    """Small per-step parameter blocks (poses, ingest parameters) go to HBM through a ring of pinned host buffers with
    asynchronous copies on the compute stream, so a step never blocks on a pageable H2D copy.  A slot is reused only
    after the copy that last read it has completed (event check).  A ring belongs to ONE stream: the device buffer of a
    slot is handed to kernels launched on the stream that was current at upload(); with ``slots`` uploads in flight on
    a second stream the oldest kernel could still be reading the buffer being overwritten.  The map classes keep one ring
    per purpose and are driven from one stream each (harness: obstacle rings on the side stream, value rings on main).  The host-side fill is a plain single-threaded
    NumPy memcpy into the pinned buffer: a torch CPU copy of >= 32 K elements would fan out over the OpenMP pool, whose
    spinning workers can exhaust a container's CPU quota and stall the whole process until the next scheduler period."""

    def __init__(self, device, nbytes: int, slots: int = 4) -> None:
        import torch

        self.device, self.nbytes = device, nbytes
        self.host = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.host_np = [h.numpy() for h in self.host]
        self.dev = [torch.empty(nbytes, dtype=torch.uint8, device=device) for _ in range(slots)]
        self.done = [None] * slots
        self.events = [None] * slots   # one interrupt-driven event per slot, reused
        self.k = 0

    def upload(self, buf):
        import torch

        if isinstance(buf, np.ndarray):
            src = np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
        else:
            src = np.frombuffer(buf, dtype=np.uint8)  # ctypes arrays / bytes expose the buffer protocol
        n = src.size
        assert n <= self.nbytes
        i = self.k
        self.k = (self.k + 1) % len(self.host)
        if self.done[i] is not None:
            self.done[i].synchronize()
        self.host_np[i][:n] = src
        self.dev[i][:n].copy_(self.host[i][:n], non_blocking=True)
        if self.events[i] is None:
            self.events[i] = torch.cuda.Event(blocking=True)
        self.events[i].record()
        self.done[i] = self.events[i]
        return self.dev[i]


# NumPy views of the C structs of include/vlfm_amd.h (filled by the library's host functions or by array ops)
VM_POSE_DTYPE = np.dtype([("inv_affine", "<f8", (6,)), ("row0", "<i4"), ("col0", "<i4"), ("env", "<i4"),
                          ("reserved", "<i4")])
INGEST_DTYPE = np.dtype([("tf", "<f8", (12,)), ("depth_scale", "<f4"), ("depth_offset", "<f4"), ("depth_max", "<f4"),
                         ("reserved0", "<f4"), ("fx", "<f8"), ("fy", "<f8"), ("min_height", "<f8"),
                         ("max_height", "<f8"), ("env", "<i4"), ("scatter", "<i4")])
assert VM_POSE_DTYPE.itemsize == 64 and INGEST_DTYPE.itemsize == 152


class _ConeTemplates:
    """Per-device cache of masked confidence templates, keyed like ValueMap._confidence_masks (value_map.py:37)."""

    def __init__(self) -> None:
        self._tmpl: Dict[Tuple[Any, ...], Any] = {}
        self._quad: Dict[Tuple[Any, ...], Any] = {}
        self._tan: Dict[Tuple[Any, ...], Any] = {}
        self._disc: Dict[Tuple[Any, ...], Any] = {}

    def template(self, device, fov: float, max_depth: float, ppm: int, min_conf: float):
        import torch

        key = (str(device), float(fov), float(max_depth), int(ppm), float(min_conf))
        if key not in self._tmpl:
            L = _lib.lib()
            size = int(max_depth * ppm)
            T = 2 * size + 1
            conf = np.zeros(T * T, np.float32)
            poly = np.zeros(2 * 512, np.int64)
            n_poly = ctypes.c_int(0)
            rc = L.vlfm_cone_template_host(float(fov), float(max_depth), int(ppm), float(min_conf),
                                           conf.ctypes.data, conf.size, poly.ctypes.data, 512, ctypes.byref(n_poly))
            _lib.check(rc, "cone_template_host")
            assert rc == T
            conf = _confidence_table_numpy(T, fov, min_conf)
            tab = conf.reshape(T, T)
            assert np.array_equal(tab, tab[::-1]) and np.array_equal(tab, tab[:, ::-1])  # |row - c|, |col - c| only
            self._quad[key] = torch.from_numpy(np.ascontiguousarray(tab[T // 2:, T // 2:])).to(device)
            d_conf = torch.from_numpy(conf).to(device)
            d_poly = torch.from_numpy(poly).to(device)
            d_tmpl = torch.empty(T * T, dtype=torch.float32, device=device)
            d_bits = torch.empty(T * ((T + 31) // 32), dtype=torch.int32, device=device)
            with torch.cuda.device(device):
                _lib.check(L.vlfm_cone_template_build(d_conf.data_ptr(), d_poly.data_ptr(), n_poly.value, T,
                                                      d_tmpl.data_ptr(), d_bits.data_ptr(), _stream_ptr()),
                           "cone_template_build")
                torch.cuda.current_stream().synchronize()
            self._tmpl[key] = (d_tmpl, d_bits, T)
        return self._tmpl[key]

    def quadrant(self, device, fov: float, max_depth: float, ppm: int, min_conf: float):
        """[(T/2+1)^2] f32: the unmasked confidence table's quadrant for the single-launch update's LDS taps."""
        self.template(device, fov, max_depth, ppm, min_conf)
        return self._quad[(str(device), float(fov), float(max_depth), int(ppm), float(min_conf))]

    def tan_table(self, device, fov: float, width: int):
        import torch

        key = (str(device), float(fov), int(width))
        if key not in self._tan:
            tab = np.zeros(width, np.float64)
            _lib.check(_lib.lib().vlfm_tan_table_host(float(fov), int(width), tab.ctypes.data), "tan_table_host")
            self._tan[key] = torch.from_numpy(tab).to(device)
        return self._tan[key]

    def disc(self, device, radius: int):
        import torch

        key = (str(device), int(radius))
        if key not in self._disc:
            hw = np.zeros(2 * radius + 1, np.int32)
            _lib.check(_lib.lib().vlfm_disc_rows_host(int(radius), hw.ctypes.data), "disc_rows_host")
            self._disc[key] = torch.from_numpy(hw).to(device)
        return self._disc[key]


_TEMPLATES = _ConeTemplates()


def _confidence_table_numpy(T: int, fov: float, min_conf: float) -> np.ndarray:
    """Unmasked confidence of value_map.py:343-351 evaluated with the reference's own math library (NumPy ufuncs for
    arctan2/cos, libm pow for the scalar ``** 2``), vectorised over the template; replaces the libm-only table of
    vlfm_cone_template_host when the host is Python so that the f64 intermediates are NumPy's to the last bit."""
    import math

    off = np.abs(np.arange(T) - T // 2)
    theta = np.arctan2(off[None, :].repeat(T, 0), off[:, None].repeat(T, 1))
    theta = (theta - 0) * (np.pi / 2 - 0) / (fov / 2 - 0) + 0
    cos = np.cos(theta)
    sq = np.array([math.pow(c, 2.0) for c in cos.reshape(-1)])
    conf = (sq - 0) * (1 - min_conf) / (1 - 0) + min_conf
    return np.ascontiguousarray(conf.astype(np.float32))


def pose_params(tf: np.ndarray, env_ids: Optional[Sequence[int]], size: int, ppm: int, T: int) -> np.ndarray:
    """Host prologue of ValueMap._localize_new_data for n observations -> structured array of vlfm_vm_pose."""
    tf = np.ascontiguousarray(np.asarray(tf, np.float64).reshape(-1, 16))
    n = tf.shape[0]
    out = np.zeros(n, VM_POSE_DTYPE)
    env = None
    if env_ids is not None:
        env = np.ascontiguousarray(np.asarray(env_ids, np.int32))
        assert env.shape == (n,)
    bad = ctypes.c_int(-1)
    # extract_yaw with the reference's own function: numpy.arctan2 (geometry_utils.py:157)
    yaw = np.ascontiguousarray(np.arctan2(tf[:, 4], tf[:, 0]))
    rc = _lib.lib().vlfm_value_map_pose_params(tf.ctypes.data, yaw.ctypes.data,
                                               env.ctypes.data if env is not None else None, n, size,
                                               ppm, T, out.ctypes.data, ctypes.byref(bad))
    _lib.check(rc, "value_map_pose_params")
    return out


class ValueMapBatch:
    """``n_envs`` value maps (+ confidence maps) resident in HBM, updated together."""

    _min_confidence: float = 0.25  # value_map.py:40

    def __init__(self, n_envs: int, value_channels: int, size: int = 1000, use_max_confidence: bool = True,
                 fusion_type: str = "default", pixels_per_meter: int = 20, device=None,
                 explored_bits: Optional[Any] = None) -> None:
        import torch

        self.device = require_gpu(device)
        _lib.lib()
        self.n_envs, self.channels, self.size, self.pixels_per_meter = n_envs, value_channels, size, pixels_per_meter
        self.use_max_confidence = bool(use_max_confidence)
        self.fusion_type = fusion_type
        if os.environ.get("MAP_FUSION_TYPE", "") != "":  # value_map.py:74-75
            self.fusion_type = os.environ["MAP_FUSION_TYPE"]
        assert self.fusion_type in _lib.FUSION_TYPES, f"Unknown fusion type {self.fusion_type}"
        self.conf = torch.zeros((n_envs, size, size), dtype=torch.float32, device=self.device)
        # f64: the reference's `_value_map` IS f64 after the first weighted fuse (value_map.py:423: `values` is an f64
        # ndarray and the attribute is re-bound to the f64 result); where the reference keeps it f32 (max-confidence :406,
        # "replace" :381-384) the kernel stores f32-representable doubles.  Either way: bit-equal to the reference.
        self.value = torch.zeros((n_envs, size, size, value_channels), dtype=torch.float64, device=self.device)
        self.n_updates = np.zeros(n_envs, np.int64)   # update_map calls per slot since construction (dtype rule below)
        # bit-packed ObstacleMap.explored_area of the same slots ([n_envs,S,ceil(S/32)] int32) when the value map is
        # synchronised with an obstacle map (value_map.py:369-375); None = Habitat default
        self.explored_bits = explored_bits
        self._colmax = None
        self._status = None
        self._ring = None
        self._written = None   # [n_envs,S,ceil(S/32)] cells that ever received a confidence (explored-synchronised mode)
        self._written_stale = False  # an update ran without the explored plane since `_written` was last complete
        self._counters = None
        self._wp_out = self._wp_host = self._wp_cells = None

    @property
    def value_is_f32(self) -> bool:
        """True when the reference's `_value_map` array never leaves f32 in this map's mode: masked assignment of
        `values` (use_max_confidence, value_map.py:406) or the "replace" ablation (:381-384, which returns before the
        weighted path).  Otherwise the first weighted fuse promotes it to f64 for good (:423)."""
        return self.fusion_type == "replace" or self.use_max_confidence

    def value_dtype(self, slot: int):
        """dtype of the reference's `_value_map` for this slot right now (f32 until the first weighted fuse; reset() keeps
        the dtype, value_map.py:96-98)."""
        return np.float32 if (self.value_is_f32 or self.n_updates[slot] == 0) else np.float64

    # ------------------------------------------------------------------------------------------ helpers
    def reset(self, env_ids: Optional[Sequence[int]] = None) -> None:
        if env_ids is None:
            self.conf.zero_()
            self.value.zero_()
            if self._written is not None:
                self._written.zero_()
        else:
            idx = list(env_ids)
            self.conf[idx] = 0
            self.value[idx] = 0
            if self._written is not None:
                self._written[idx] = 0

    def _rings(self, n: int):
        if self._ring is None or self._ring.nbytes < max(n, self.n_envs) * 256:
            self._ring = UploadRing(self.device, max(n, self.n_envs) * 256, slots=8)
        return self._ring

    def _scratch(self, n: int, width: int):
        import torch

        if self._colmax is None or self._colmax.shape[0] < n or self._colmax.shape[1] != width:
            # order-preserving u32 keys, zero = "-inf"; produced by depth ingest, consumed + re-zeroed by the update
            self._colmax = torch.zeros((max(n, self.n_envs), width), dtype=torch.int32, device=self.device)
            self._status = torch.zeros((max(n, self.n_envs), 2), dtype=torch.int32, device=self.device)
        return self._colmax, self._status

    def column_max(self, depth) -> Any:
        """np.max(depth, axis=0) for a [n,H,W] device tensor via the depth-ingest kernel (no obstacle scatter).
        Returns the column-max KEY buffer that update() consumes."""
        import torch

        n, H, W = depth.shape
        colmax, status = self._scratch(n, W)
        prm = np.zeros(n, INGEST_DTYPE)
        with torch.cuda.device(self.device):
            d_prm = self._rings(n).upload(prm)
            _lib.check(_lib.lib().vlfm_depth_ingest_batched(depth.data_ptr(), n, H, W, d_prm.data_ptr(),
                                                           colmax.data_ptr(), None, self.size, self.pixels_per_meter,
                                                           status.data_ptr(), None, None, None, _stream_ptr()),
                       "depth_ingest")
        return colmax[:n]

    # ------------------------------------------------------------------------------------------ update
    def update(self, values, depth, tf_camera_to_episodic, min_depth: float, max_depth: float, fov: float,
               env_ids: Optional[Sequence[int]] = None, colmax=None) -> None:
        """ValueMap.update_map for n observations at once (value_map.py:100-128).

        values [n,C] f64 host array or device tensor; depth [n,H,W] f32 (device tensor, or host array that is
        uploaded); tf [n,4,4] f64 host; ``colmax`` = column-max keys [n,W] on device (from a shared depth ingest).
        """
        import torch

        if torch.is_tensor(values):  # device-resident scores straight from the ITC head: no host round trip
            d_vals = values.to(device=self.device, dtype=torch.float64).reshape(-1, self.channels).contiguous()
        else:
            d_vals = torch.from_numpy(
                np.ascontiguousarray(np.asarray(values, np.float64).reshape(-1, self.channels))).to(self.device)
        n = d_vals.shape[0]
        # Host prologue FIRST: a camera outside the map (AssertionError, img_utils.py:43) or a bad slot list must fail before
        # any column maximum is reduced into the key buffer -- keys are only re-zeroed by a completed update, and stale keys
        # would max-merge this frame's depth profile into the next one.  Keys handed in by a shared depth ingest are
        # zeroed on that error path for the same reason.
        try:
            W = colmax.shape[-1] if colmax is not None else int(depth.shape[-1])
            d_tmpl, d_bits, T = _TEMPLATES.template(self.device, fov, max_depth, self.pixels_per_meter,
                                                    self._min_confidence)
            d_tan = _TEMPLATES.tan_table(self.device, fov, W)
            pose = pose_params(tf_camera_to_episodic, env_ids, self.size, self.pixels_per_meter, T)
            # one launch fuses every observation concurrently: two observations of the SAME slot (the reference's
            # multi-camera loop, reality_policies.py:113-141, is sequential) must go through separate calls
            assert len(np.unique(pose["env"])) == len(pose), "one observation per environment slot and call"
            assert int(pose["env"].max()) < self.n_envs and int(pose["env"].min()) >= 0, "environment slot out of range"
            self.n_updates[pose["env"]] += 1
        except Exception:
            if colmax is not None:
                colmax.zero_()
            raise
        if colmax is None:
            if not torch.is_tensor(depth):
                depth = torch.from_numpy(np.ascontiguousarray(depth, np.float32)).to(self.device)
            depth = depth.reshape(n, depth.shape[-2], depth.shape[-1]).contiguous()
            assert depth.dtype == torch.float32
            colmax = self.column_max(depth)
        L = _lib.lib()
        with torch.cuda.device(self.device):
            ring = self._rings(n)
            d_pose = ring.upload(pose)
            explored_ptr = None
            if self.explored_bits is not None:
                ex = self.explored_bits
                assert ex.dtype == torch.int32 and ex.is_contiguous() and ex.shape[-2] == self.size
                explored_ptr = ex.data_ptr()
            written_ptr = None
            if explored_ptr is None:
                self._written_stale = True   # cells fused from here on are not recorded in the plane
            else:
                if self._written is None or self._written_stale:
                    # conf may already hold values (an obstacle map attached mid-episode, or detached for a while and
                    # attached again): (re)start from conf != 0
                    if self._written is None:
                        self._written = torch.zeros((self.n_envs, self.size, (self.size + 31) // 32),
                                                    dtype=torch.int32, device=self.device)
                    _lib.check(L.vlfm_bits_pack((self.conf != 0).to(torch.uint8).contiguous().data_ptr(),
                                                self._written.data_ptr(), self.n_envs, self.size, self.size,
                                                _stream_ptr()), "bits_pack")
                    self._written_stale = False
                written_ptr = self._written.data_ptr()
            if self._counters is None or self._counters.numel() < n:
                self._counters = torch.zeros(max(n, self.n_envs), dtype=torch.int32, device=self.device)
            _lib.check(L.vlfm_value_map_update_fused_batched(
                colmax.data_ptr(), W, d_tan.data_ptr(), d_tmpl.data_ptr(), d_bits.data_ptr(), T, d_pose.data_ptr(),
                d_vals.data_ptr(), n, self.conf.data_ptr(), self.value.data_ptr(), self.size, self.channels,
                self.pixels_per_meter, float(min_depth), float(max_depth), int(self.use_max_confidence),
                _lib.FUSION_TYPES[self.fusion_type], explored_ptr, written_ptr, self._counters.data_ptr(),
                _TEMPLATES.quadrant(self.device, fov, max_depth, self.pixels_per_meter,
                                    self._min_confidence).data_ptr(), _stream_ptr()), "value_map_update_fused")

    # ------------------------------------------------------------------------------------------ frontier scoring
    def waypoint_values(self, waypoints_xy: np.ndarray, env_of_waypoint: Sequence[int], radius: float) -> np.ndarray:
        """Disc medians for m waypoints -> [m, C] f64 (value_map.py:161-176, img_utils.py:213-266): np.median's result in
        the dtype of the reference's array (f64 in the weighted mode, f32 -- widened exactly -- otherwise)."""
        import torch

        wp = np.asarray(waypoints_xy, np.float64).reshape(-1, 2)
        m = wp.shape[0]
        if m == 0:
            return np.zeros((0, self.channels), np.float64)
        radius_px = int(radius * self.pixels_per_meter)
        # int() truncates toward zero (value_map.py:165-166) == ndarray.astype(int64) for finite values
        px = (-wp[:, 0] * self.pixels_per_meter).astype(np.int64) + self.size // 2
        py = (-wp[:, 1] * self.pixels_per_meter).astype(np.int64) + self.size // 2
        row, col = self.size - px, py
        assert ((row >= 0) & (row < self.size) & (col >= 0) & (col < self.size)).all(), \
            "Pixel location is outside the image."
        cells = np.stack([np.asarray(env_of_waypoint, np.int64), row, col], axis=1).astype(np.int32)
        d_disc = _TEMPLATES.disc(self.device, radius_px)
        if self._wp_out is None or self._wp_out.shape[0] < m:
            cap = max(64, 1 << (m - 1).bit_length())
            self._wp_out = torch.empty((cap, self.channels), dtype=torch.float64, device=self.device)
            self._wp_host = torch.empty((cap, self.channels), dtype=torch.float64).pin_memory()
            self._wp_cells = UploadRing(self.device, cap * 12, slots=4)
        with torch.cuda.device(self.device):
            d_cells = self._wp_cells.upload(cells)
            _lib.check(_lib.lib().vlfm_value_map_sort_waypoints_batched(self.value.data_ptr(), self.size,
                                                                       self.channels, d_cells.data_ptr(), m,
                                                                       radius_px, d_disc.data_ptr(),
                                                                       int(self.value_is_f32), self._wp_out.data_ptr(),
                                                                       _stream_ptr()),
                       "sort_waypoints")
            self._wp_host[:m].copy_(self._wp_out[:m], non_blocking=True)
            wait_stream()
        return self._wp_host[:m].numpy().copy()


def _explored_bits_of(obstacle_map, device):
    """Bit-packed explored area [1,S,ceil(S/32)] of an attached obstacle map: ours hands over its HBM plane; any other
    object exposing the reference attribute ``explored_area`` (S,S bool) is uploaded and packed on the device."""
    import torch

    if hasattr(obstacle_map, "explored_bits_device"):
        return obstacle_map.explored_bits_device()
    area = np.ascontiguousarray(np.asarray(obstacle_map.explored_area).astype(np.uint8))
    S = area.shape[0]
    src = torch.from_numpy(area).to(device)
    dst = torch.empty((1, S, (S + 31) // 32), dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().vlfm_bits_pack(src.data_ptr(), dst.data_ptr(), 1, S, S, _stream_ptr()), "bits_pack")
    return dst


# value_map.py:26-30 -- the reference's record/replay switches and on-disk format (8-bit depth PNGs + data.json + kwargs.json)
RECORDING = os.environ.get("RECORD_VALUE_MAP", "0") == "1"
PLAYING = os.environ.get("PLAY_VALUE_MAP", "0") == "1"
RECORDING_DIR = "value_map_recordings"
JSON_PATH = os.path.join(RECORDING_DIR, "data.json")
KWARGS_JSON = os.path.join(RECORDING_DIR, "kwargs.json")


class ValueMap(BaseMap):
    """Drop-in for vlfm.mapping.value_map.ValueMap (same constructor and method signatures)."""

    _confidence_masks: Dict[Tuple[float, float], np.ndarray] = {}
    _min_confidence: float = 0.25
    _decision_threshold: float = 0.35

    def __init__(self, value_channels: int, size: int = 1000, use_max_confidence: bool = True,
                 fusion_type: str = "default", obstacle_map: Optional["ObstacleMap"] = None,  # noqa: F821
                 device=None, _batch: Optional[ValueMapBatch] = None, _slot: int = 0) -> None:
        super().__init__(size)
        self._value_channels = value_channels
        self._use_max_confidence = use_max_confidence
        self._obstacle_map = obstacle_map
        if obstacle_map is not None:
            assert obstacle_map.pixels_per_meter == self.pixels_per_meter  # value_map.py:71-73
            assert obstacle_map.size == self.size
        if _batch is None:
            _batch = ValueMapBatch(1, value_channels, size, use_max_confidence, fusion_type, self.pixels_per_meter,
                                   device)
            _slot = 0
        self._batch, self._slot = _batch, _slot
        self._fusion_type = _batch.fusion_type
        if RECORDING:  # value_map.py:77-94
            import json
            import shutil
            import warnings

            if os.path.isdir(RECORDING_DIR):
                warnings.warn(f"Recording directory {RECORDING_DIR} already exists. Deleting it.")
                shutil.rmtree(RECORDING_DIR)
            os.mkdir(RECORDING_DIR)
            with open(KWARGS_JSON, "w") as f:
                json.dump({"value_channels": value_channels, "size": size, "use_max_confidence": use_max_confidence}, f)
            with open(JSON_PATH, "w") as f:
                f.write("{}")

    # maps are HBM-resident; these attributes are host snapshots for the callers that read them (visualisation)
    @property
    def _map(self) -> np.ndarray:
        return self._batch.conf[self._slot].cpu().numpy()

    @property
    def _value_map(self) -> np.ndarray:
        # the reference's array is f32 until the first weighted fuse and f64 from then on (value_map.py:423); in the modes
        # that assign `values` into it (max-confidence, "replace") it stays f32.  The stored doubles are exact either way.
        return self._batch.value[self._slot].cpu().numpy().astype(self._batch.value_dtype(self._slot), copy=False)

    def reset(self) -> None:
        super().reset()
        self._batch.reset([self._slot])

    def update_map(self, values: np.ndarray, depth: np.ndarray, tf_camera_to_episodic: np.ndarray,
                   min_depth: float, max_depth: float, fov: float) -> None:
        assert len(values) == self._value_channels, \
            f"Incorrect number of values given ({len(values)}). Expected {self._value_channels}."
        import torch

        if not torch.is_tensor(depth):
            depth = np.asarray(depth, np.float32)
            if depth.ndim == 3:
                depth = depth.squeeze(2)  # value_map.py:231-232
            depth = depth[None]
        else:
            depth = depth.reshape(1, depth.shape[0], depth.shape[1])
        if self._obstacle_map is not None:
            self._batch.explored_bits = _explored_bits_of(self._obstacle_map, self._batch.device)
        self._batch.update(np.asarray(values, np.float64)[None], depth, np.asarray(tf_camera_to_episodic)[None],
                           min_depth, max_depth, fov, env_ids=[self._slot])
        if RECORDING:  # value_map.py:130-144 (cv2.imwrite of a single-channel u8 image == an 8-bit greyscale PNG)
            import glob
            import json

            from PIL import Image

            d = depth.cpu().numpy() if torch.is_tensor(depth) else np.asarray(depth)
            idx = len(glob.glob(os.path.join(RECORDING_DIR, "*.png")))
            img_path = os.path.join(RECORDING_DIR, f"{idx:04d}.png")
            Image.fromarray((d.reshape(d.shape[-2], d.shape[-1]) * 255).astype(np.uint8)).save(img_path)
            with open(JSON_PATH, "r") as f:
                data = json.load(f)
            data[img_path] = {"values": np.asarray(values).tolist(),
                              "tf_camera_to_episodic": np.asarray(tf_camera_to_episodic).tolist(),
                              "min_depth": min_depth, "max_depth": max_depth, "fov": fov}
            with open(JSON_PATH, "w") as f:
                json.dump(data, f)

    def sort_waypoints(self, waypoints: np.ndarray, radius: float,
                       reduce_fn: Optional[Callable] = None) -> Tuple[np.ndarray, List[float]]:
        vals = self._batch.waypoint_values(waypoints, [self._slot] * len(waypoints), radius)
        if self._value_channels == 1:
            values: List[Any] = [float(v) if v != -1 else -1 for v in vals[:, 0]]
        else:
            values = [tuple(float(c) for c in v) for v in vals]
            assert reduce_fn is not None, "Must provide a reduction function when using multiple value channels."
            values = reduce_fn(values)
        order = np.argsort([-v for v in values])  # the reference's own ordering call (value_map.py:183)
        return np.array([waypoints[i] for i in order]), [values[i] for i in order]

    def visualize(self, markers=None, reduce_fn: Callable = lambda i: np.max(i, axis=-1), obstacle_map=None):
        """Plain grey-scale rendering (the reference's OpenCV inferno/trajectory drawing is out of scope)."""
        reduced = reduce_fn(self._value_map).copy()
        if obstacle_map is not None:
            reduced[obstacle_map.explored_area == 0] = 0
        img = np.flipud(reduced)
        zero = img == 0
        top = float(img.max()) if img.size else 0.0
        scaled = np.zeros(img.shape, np.uint8) if top <= 0 else (img / top * 255).astype(np.uint8)
        rgb = np.stack([scaled] * 3, axis=-1)
        rgb[zero] = (255, 255, 255)
        return rgb


def replay_from_dir(recording_dir: str = RECORDING_DIR, device=None) -> ValueMap:
    """value_map.py:448-475 without the interactive window: rebuilds a ValueMap from a recording made by the reference
    (or by this class) and returns it.  ``PLAY_VALUE_MAP=1 python -m vlfm_amd.mapping.value_map`` mirrors the reference's
    entry point and writes ``value_map_replay.png``."""
    import json

    from PIL import Image

    with open(os.path.join(recording_dir, "kwargs.json"), "r") as f:
        kwargs = json.load(f)
    with open(os.path.join(recording_dir, "data.json"), "r") as f:
        data = json.load(f)
    v = ValueMap(device=device, **kwargs)
    for img_path in sorted(data.keys()):
        rec = data[img_path]
        path = img_path if os.path.exists(img_path) else os.path.join(recording_dir, os.path.basename(img_path))
        depth = np.asarray(Image.open(path).convert("L")).astype(np.float32) / 255.0
        v.update_map(np.array(rec["values"]), depth, np.array(rec["tf_camera_to_episodic"]), float(rec["min_depth"]),
                     float(rec["max_depth"]), float(rec["fov"]))
    return v


if __name__ == "__main__":
    if PLAYING:
        from PIL import Image

        Image.fromarray(replay_from_dir().visualize()).save("value_map_replay.png")
        print("wrote value_map_replay.png")

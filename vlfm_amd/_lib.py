"""ctypes binding of libvlfm_amd.so (the C ABI declared in include/vlfm_amd.h).

There is deliberately NO fallback: if the shared library is missing the import of any map class fails
loudly, and if no HIP device is present the device entry points are never reached because the map
classes refuse to construct.  The oracle under oracle/ is test infrastructure and is never imported here.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLFM_LIB_PATH") or os.path.join(_HERE, "libvlfm_amd.so")  # override: diagnostic builds
SOURCES = ["value_map.hip", "depth_ingest.hip", "depth_holes.hip", "obstacle_map.hip", "vlm_ops.hip", "vit_attention.hip", "detect_ops.hip", "object_cloud.hip", "gemm_f16.hip", "gemm_f32.hip", "conv_nhwc.hip", "sam_ops.hip", "host.cpp"]

VLFM_OK = 0
VLFM_ERR_INVALID = -1
VLFM_ERR_OUTSIDE_MAP = -2
VLFM_ERR_INDEX = -3
VLFM_ERR_HIP = -4
VLFM_ERR_CAPACITY = -5

FUSION_TYPES = {"default": 0, "replace": 1, "equal_weighting": 2}


class VmPose(ctypes.Structure):
    _fields_ = [("inv_affine", ctypes.c_double * 6), ("row0", ctypes.c_int32), ("col0", ctypes.c_int32),
                ("env", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class IngestParams(ctypes.Structure):
    _fields_ = [("tf", ctypes.c_double * 12), ("depth_scale", ctypes.c_float), ("depth_offset", ctypes.c_float),
                ("depth_max", ctypes.c_float), ("reserved0", ctypes.c_float), ("fx", ctypes.c_double),
                ("fy", ctypes.c_double), ("min_height", ctypes.c_double), ("max_height", ctypes.c_double),
                ("env", ctypes.c_int32), ("scatter", ctypes.c_int32)]


FOG_MAX_POLY = 80


class FogParams(ctypes.Structure):
    _fields_ = [("env", ctypes.c_int32), ("ax", ctypes.c_int32), ("ay", ctypes.c_int32), ("radius", ctypes.c_int32),
                ("n_poly", ctypes.c_int32), ("reserved", ctypes.c_int32), ("rot_c", ctypes.c_double),
                ("rot_s", ctypes.c_double), ("line_len", ctypes.c_double), ("poly", ctypes.c_longlong * (2 * FOG_MAX_POLY))]


class ScatterJournal(ctypes.Structure):
    _fields_ = [("d_cells", ctypes.c_void_p), ("d_count", ctypes.c_void_p), ("capacity", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


assert ctypes.sizeof(ScatterJournal) == 24
assert ctypes.sizeof(VmPose) == 64 and ctypes.sizeof(IngestParams) == 152 and ctypes.sizeof(FogParams) == 48 + 16 * FOG_MAX_POLY


def build(verbose: bool = False) -> str:
    """Compile every HIP/C++ source under vlfm_amd/csrc for gfx950 into vlfm_amd/libvlfm_amd.so (in-tree).  One object
    per source (rebuilt only when the source or a header changed, compiled in parallel), then one link."""
    from concurrent.futures import ThreadPoolExecutor

    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, s) for s in SOURCES if os.path.exists(os.path.join(csrc, s))]
    hdrs = [os.path.join(csrc, h) for h in os.listdir(csrc) if h.endswith(".h")] + [
        os.path.join(_HERE, "..", "include", "vlfm_amd.h")]
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    objdir = os.path.join(csrc, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off"]

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            cmd = ["hipcc"] + flags + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as pool:
        objs = list(pool.map(compile_one, srcs))
    if os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB_PATH


_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  vlfm_amd has no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.vlfm_last_error.restype = ctypes.c_char_p
        L.vlfm_abi_version.restype = ci
        L.vlfm_profile_enable.argtypes = [ci]
        L.vlfm_host_wait_mode.argtypes = [ci]
        L.vlfm_profile_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(cd), ctypes.POINTER(ci)]
        L.vlfm_value_map_pose_params.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp, ctypes.POINTER(ci)]
        L.vlfm_cone_template_host.argtypes = [cd, cd, ci, cd, vp, ci, vp, ci, ctypes.POINTER(ci)]
        L.vlfm_tan_table_host.argtypes = [cd, ci, vp]
        L.vlfm_disc_rows_host.argtypes = [ci, vp]
        L.vlfm_cone_template_build.argtypes = [vp, vp, ci, ci, vp, vp, vp]
        L.vlfm_depth_ingest_batched.argtypes = [vp, ci, ci, ci, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp]
        L.vlfm_selftest_div_exact.argtypes = [vp, ci, cd, vp, vp]
        L.vlfm_hole_scratch_bytes.argtypes = [ci, ci, ci, ci, ci]
        L.vlfm_hole_scratch_bytes.restype = ctypes.c_size_t
        L.vlfm_fill_small_holes_batched.argtypes = [vp, vp, ci, ci, ci, cd, vp, ctypes.c_size_t, ci, ci, vp, vp, vp, vp,
                                                    ci, vp, vp]
        L.vlfm_depth_scatter_holes_batched.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, ci, ci, vp, vp, vp]
        L.vlfm_layernorm_bias_f16.argtypes = [vp, vp, vp, vp, vp, ci, ci, ctypes.c_float, vp]
        L.vlfm_vit_attention_f16.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_float, vp]
        L.vlfm_gemm_f16_nt.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
        L.vlfm_gemm_f32_nt.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp]
        L.vlfm_split_f32_to_f16_pair.argtypes = [vp, vp, vp, ctypes.c_longlong, vp, vp]
        L.vlfm_layernorm_rows_f32.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ctypes.c_float, vp]
        L.vlfm_window_reverse_add_f32.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
        L.vlfm_dwconv3x3_nhwc_f32.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, vp]
        L.vlfm_window_attention_f32.argtypes = [vp, vp, vp, ctypes.c_longlong, ci, ci, ctypes.c_float, vp]
        L.vlfm_layernorm_rows_shifted_f32.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ctypes.c_float, ci, ci, vp]
        L.vlfm_window_reverse_add_shifted_f32.argtypes = [vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.vlfm_window_attention_masked_f32.argtypes = [vp, vp, vp, ci, vp, ctypes.c_longlong, ci, ci, ctypes.c_float, vp]
        L.vlfm_maxpool2x2_nhwc_f16.argtypes = [vp, vp, ci, ci, ci, ci, vp]
        L.vlfm_conv_nhwc_tile.argtypes = [ci, ci, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci)]
        L.vlfm_conv_nhwc_f16.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, ci, ci, vp]
        L.vlfm_value_map_update_fused_batched.argtypes = [vp, ci, vp, vp, vp, ci, vp, vp, ci, vp, vp, ci, ci, ci, cd, cd,
                                                          ci, ci, vp, vp, vp, vp, vp]
        L.vlfm_value_map_sort_waypoints_batched.argtypes = [vp, ci, ci, vp, ci, ci, vp, ci, vp, vp]
        L.vlfm_resample_coeffs_host.argtypes = [ci, ci, vp, vp, ci, ctypes.POINTER(ci)]
        L.vlfm_resample_coeffs_filter_host.argtypes = [ci, ci, ci, vp, vp, ci, ctypes.POINTER(ci)]
        L.vlfm_preprocess_sam_batched.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, ci, vp, vp, ci, vp, vp, ci, vp, vp, vp]
        L.vlfm_preprocess_rgb_batched.argtypes = [vp, ci, ci, ci, ci, vp, vp, ci, vp, vp, ci, vp, vp, vp, vp, ci, ci, vp]
        L.vlfm_itc_head_batched.argtypes = [vp, ci, ci, ci, vp, vp, vp]
        cf = ctypes.c_float
        L.vlfm_resize_area_tab_host.argtypes = [ci, ci, vp, vp, vp, ci]
        L.vlfm_resize_area_batched.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, ci, vp, ci, vp]
        L.vlfm_to_tensor_normalize_batched.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp]
        L.vlfm_nms_scratch_bytes.argtypes = [ci]
        L.vlfm_nms_scratch_bytes.restype = ctypes.c_size_t
        L.vlfm_nms.argtypes = [vp, vp, ci, cf, vp, ctypes.c_size_t, vp, vp, ci, vp]
        L.vlfm_ms_deform_attn.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp, vp]
        L.vlfm_ms_deform_attn_fused.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, ci, vp, vp]
        L.vlfm_object_cloud_scratch_bytes.argtypes = [ci, ci]
        L.vlfm_object_cloud_scratch_bytes.restype = ctypes.c_size_t
        L.vlfm_object_cloud_extract.argtypes = [vp, vp, ci, ci, ci, cd, cd, cd, cd, vp, vp, ci, vp, vp]
        L.vlfm_dbscan_scratch_bytes.argtypes = [ci]
        L.vlfm_dbscan_scratch_bytes.restype = ctypes.c_size_t
        L.vlfm_dbscan_largest_cluster.argtypes = [vp, ci, cd, ci, vp, ctypes.c_size_t, vp, vp, vp, vp]
        L.vlfm_bits_pack.argtypes = [vp, vp, ci, ci, ci, vp]
        L.vlfm_bits_unpack.argtypes = [vp, vp, ci, ci, ci, vp]
        L.vlfm_bits_dilate.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp]
        L.vlfm_find_contours_external.argtypes = [vp, ci, ci, ci, ci, vp, vp, ci, vp, vp, ci, vp, vp]
        L.vlfm_dwconv3x3_f32.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
        L.vlfm_bias_act_nchw.argtypes = [vp, vp, ci, ci, ctypes.c_longlong, ci, ci, vp]
        L.vlfm_find_contours_wg_scratch_bytes.argtypes = [ci, ci, ci, ci]
        L.vlfm_find_contours_wg_scratch_bytes.restype = ctypes.c_size_t
        L.vlfm_find_contours_external_wg.argtypes = [vp, ci, ci, ci, ci, vp, ctypes.c_size_t, vp, ci, vp, vp, ci, vp, vp]
        L.vlfm_walk_path_counters.argtypes = [vp, ci]
        L.vlfm_fog_params_host.argtypes = [vp, vp, vp, cd, cd, vp, vp, ci, vp]
        L.vlfm_obstacle_scratch_bytes.argtypes = [ci, ci, ci, ci]
        L.vlfm_obstacle_scratch_bytes.restype = ctypes.c_size_t
        L.vlfm_obstacle_map_update_batched.argtypes = [vp, ci, vp, vp, vp, vp, ci, ci, ci, ci, cd, vp, ctypes.c_size_t,
                                                       ci, ci, vp, ci, vp, ci, ci, vp, ci, ci, vp]
        L.vlfm_obstacle_status.argtypes = [vp, ci, ci, ci, ci, vp]
        _lib = L
    return _lib


def profile_read(kernel: str):
    """(mean_ms, launches) of a kernel since vlfm_profile_enable(1)."""
    ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
    lib().vlfm_profile_read(kernel.encode(), ctypes.byref(ms), ctypes.byref(n))
    return ms.value, n.value


def last_error() -> str:
    return lib().vlfm_last_error().decode()


def host_wait_blocking(device=None, blocking: bool = True) -> None:
    """Make host waits on `device` (default: the current one) sleep on the completion interrupt instead of spinning
    (vlfm_host_wait_mode).  Call it before the device's first stream / allocation: the runtime fixes a queue's wait
    policy when it creates the queue (measured: set after the model was built, the waiting thread still spun)."""
    import torch

    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(lib().vlfm_host_wait_mode(int(bool(blocking))), "host_wait_mode")


def try_host_wait_blocking(device=None) -> bool:
    """`host_wait_blocking` for library code that does not own the process: the flag changes how EVERY wait on the device behaves
    and some runtimes reject a change once the primary context is active, so a failure is a warning, never an exception, and
    VLFM_HOST_WAIT=spin opts out.  Returns whether the flag was set."""
    import os
    import warnings

    if os.environ.get("VLFM_HOST_WAIT", "blocking") == "spin":
        return False
    try:
        host_wait_blocking(device)
        return True
    except Exception as exc:  # noqa: BLE001
        warnings.warn(f"blocking host waits not enabled on {device}: {exc} (call vlfm_amd._lib.host_wait_blocking before the "
                      f"device's first allocation; waits will spin on a host core)")
        return False


def check(rc: int, what: str = "") -> int:
    """Map a vlfm_status to the reference's Python exception conventions."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == VLFM_ERR_OUTSIDE_MAP:
        raise AssertionError(msg or "Pixel location is outside the image.")
    if rc == VLFM_ERR_INDEX:
        raise IndexError(msg or "index out of bounds for obstacle map")
    raise RuntimeError(f"libvlfm_amd {what} failed ({rc}): {msg}")

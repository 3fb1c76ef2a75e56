"""oracle/ref_shim.py -- TEST INFRASTRUCTURE ONLY.

Makes the REFERENCE'S OWN Python sources importable in this container, so that the reference itself (not a
restatement) generates the golden vectors under tests/golden/ and cross-checks the oracle:

    /root/reference/vlfm/mapping/{base_map,value_map,obstacle_map}.py
    /root/reference/vlfm/utils/{geometry_utils,img_utils}.py

Those modules import three packages that are absent here and not installable (no network):
``cv2`` (opencv-python==4.5.5.64), ``frontier_exploration`` (un-pinned git HEAD) and -- only for visualisation -- a few
more cv2 entry points.  install() registers stand-in modules for exactly those names:

    cv2                                   -> oracle/cv.py  (facade over oracle/cvport.c, OpenCV 4.5.5 restated)
    frontier_exploration.frontier_detection.detect_frontier_waypoints
    frontier_exploration.utils.fog_of_war.reveal_fog_of_war
                                          -> oracle/ref_frontier_exploration.py (restated by contract)

What this pins: every line of arithmetic that lives IN the reference tree (dtype promotions, truncation vs rint, the
fusion algebra, the depth-profile polygon, index conventions, control flow, error behaviour) is executed from the
reference's own source.  What it does NOT pin: the behaviour of the stand-ins themselves (OpenCV rasterisation rules,
frontier_exploration) -- those remain restatements checked only by the hand-derived known answers in
tests/test_oracle_cv.py.  /root/reference does not exist on the GPU box: only the generator script
(tests/golden/make_golden.py) and the live cross-check in tests/test_golden_reference.py (skipped when the reference
is absent) call install().

REAL LIBRARIES: with ``VLFM_REAL_CV2=1`` in the environment (or ``install(real=True)``) nothing is planted for
a package that imports for real: ``cv2`` MUST then be the real one (ImportError otherwise), ``frontier_exploration`` /
``torchvision`` / ``open3d`` are used when importable and stood in for otherwise; ``backends()`` says which is which.  That
turns every fixture check into "the reference on real OpenCV vs the committed fixture" on any machine with
``opencv-python==4.5.5.64`` (tools/verify_with_real_vlfm.md).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


_backends: dict = {}


def backends() -> dict:
    """{package: "real <version> (<file>)" | "stand-in: ..."} of the last install()."""
    return dict(_backends)


def _real_or_none(name: str):
    """The real top-level package ``name`` if it imports (stand-ins carry ``__vlfm_standin__``), else None."""
    import importlib

    have = sys.modules.get(name)
    if have is not None and getattr(have, "__vlfm_standin__", False):
        for m in [m for m in sys.modules if m.split(".")[0] == name]:
            del sys.modules[m]
    try:
        mod = importlib.import_module(name)
    except Exception:  # noqa: BLE001 - ImportError, or a broken binary wheel
        return None
    return None if getattr(mod, "__vlfm_standin__", False) else mod


def _mark(*mods) -> None:
    for m in mods:
        m.__vlfm_standin__ = True


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "vlfm", "mapping", "value_map.py"))


_PLANTED = ("cv2", "frontier_exploration", "torchvision", "open3d", "hydra", "vlfm")
_before: dict = {}


def uninstall() -> None:
    """Remove every stand-in (and the reference's modules) from sys.modules again and restore what was there before, so
    that code imported later in the same process (transformers probes for torchvision!) never sees a fake package."""
    if not _before.get("installed"):
        return
    for name in [m for m in sys.modules if m.split(".")[0] in _PLANTED]:
        del sys.modules[name]
    sys.modules.update(_before.get("modules", {}))
    if REFERENCE_ROOT in sys.path and not _before.get("path_had_root"):
        sys.path.remove(REFERENCE_ROOT)
    _before.clear()


def install(real=None) -> None:
    """Idempotent.  Plants fake top-level modules in sys.modules: pair with uninstall() (tests/conftest.py does, after every
    test) or call in a dedicated process.  ``real`` (default: the VLFM_REAL_CV2 environment switch): use the real packages
    where they import -- cv2 is then mandatory."""
    from . import cv as facade

    real = facade.real_cv2_requested() if real is None else bool(real)
    if "vlfm.mapping.value_map" in sys.modules and _before.get("installed") and _before.get("real") == real:
        return
    if not _before.get("installed"):
        _before.update(installed=True, path_had_root=REFERENCE_ROOT in sys.path,
                       modules={m: v for m, v in sys.modules.items() if m.split(".")[0] in _PLANTED})
    if not available():
        raise ImportError(f"{REFERENCE_ROOT} is not present (GPU box?) -- golden fixtures are the pin there")
    from . import ref_frontier_exploration as fe

    _before["real"] = real
    _backends.clear()
    if real:
        cv2_real = facade.import_real_cv2()     # loud: VLFM_REAL_CV2=1 without an importable cv2 is an error, not a fallback
        sys.modules["cv2"] = cv2_real
        _backends["cv2"] = f"real {getattr(cv2_real, '__version__', '?')} ({getattr(cv2_real, '__file__', '?')})"
        fe_real = _real_or_none("frontier_exploration")
        _backends["frontier_exploration"] = (f"real ({getattr(fe_real, '__file__', '?')})" if fe_real is not None
                                             else "stand-in: oracle/ref_frontier_exploration.py")
    else:
        cv2_real = fe_real = None
        _backends["cv2"] = "stand-in: oracle/cvport.c"
        _backends["frontier_exploration"] = "stand-in: oracle/ref_frontier_exploration.py"
    _plant_cv2_and_frontier(facade, fe, plant_cv2=cv2_real is None, plant_fe=fe_real is None)
    _plant_rest(real)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def _plant_cv2_and_frontier(facade, fe, plant_cv2: bool, plant_fe: bool) -> None:
    if not plant_cv2:
        return _plant_frontier(fe) if plant_fe else None
    cv2 = types.ModuleType("cv2")
    _mark(cv2)
    cv2.__doc__ = "stand-in for opencv-python==4.5.5.64 backed by oracle/cvport.c (see oracle/ref_shim.py)"
    for name in ("ellipse", "drawContours", "circle", "polylines", "getRotationMatrix2D", "warpAffine", "dilate",
                 "blur", "findContours", "contourArea", "pointPolygonTest", "isContourConvex", "bitwise_and", "erode",
                 "boundingRect",
                 "RETR_EXTERNAL", "RETR_LIST", "RETR_CCOMP", "RETR_TREE", "CHAIN_APPROX_NONE", "CHAIN_APPROX_SIMPLE"):
        setattr(cv2, name, getattr(facade, name))
    # constants that visualisation-only code paths name (never evaluated by the golden generator)
    for k, name in enumerate(("COLORMAP_INFERNO", "COLOR_BGR2RGB", "COLOR_GRAY2RGB", "COLOR_GRAY2BGR", "INTER_AREA",
                              "BORDER_CONSTANT", "IMREAD_GRAYSCALE", "FONT_HERSHEY_SIMPLEX")):
        setattr(cv2, name, k)
    sys.modules["cv2"] = cv2
    if plant_fe:
        _plant_frontier(fe)


def _plant_frontier(fe) -> None:
    pkg = types.ModuleType("frontier_exploration")
    pkg.__path__ = []  # mark as package
    det = types.ModuleType("frontier_exploration.frontier_detection")
    det.detect_frontier_waypoints = fe.detect_frontier_waypoints
    utils = types.ModuleType("frontier_exploration.utils")
    utils.__path__ = []
    fow = types.ModuleType("frontier_exploration.utils.fog_of_war")
    fow.reveal_fog_of_war = fe.reveal_fog_of_war
    pkg.frontier_detection, pkg.utils, utils.fog_of_war = det, utils, fow
    _mark(pkg, det, utils, fow)
    sys.modules.update({"frontier_exploration": pkg, "frontier_exploration.frontier_detection": det,
                        "frontier_exploration.utils": utils, "frontier_exploration.utils.fog_of_war": fow})


def _plant_rest(real: bool) -> None:
    # torchvision (==0.13.1 in the reference) is absent: only ops.box_convert is touched at import/use time by
    # vlfm/vlm/detections.py:10,31.  Stand-in = torchvision's published formula, written independently of the product's.
    import torch

    def _box_convert(boxes, in_fmt, out_fmt):
        assert in_fmt == "cxcywh" and out_fmt == "xyxy"
        cx, cy, w, h = boxes.unbind(-1)
        return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)

    if real and _real_or_none("torchvision") is not None:
        _backends["torchvision"] = "real"
    else:
        _backends["torchvision"] = "stand-in: box_convert only"
    tv = types.ModuleType("torchvision")
    tv.__path__ = []
    tv_ops = types.ModuleType("torchvision.ops")
    tv_ops.box_convert = _box_convert
    tv.ops = tv_ops
    # grounding_dino.py:7 imports torchvision.transforms.functional at module level (only used inside the model server)
    tv_tf = types.ModuleType("torchvision.transforms")
    tv_tf.__path__ = []
    tv_tf.functional = types.ModuleType("torchvision.transforms.functional")
    tv.transforms = tv_tf
    _mark(tv, tv_ops, tv_tf, tv_tf.functional)
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.ops", tv_ops)
    sys.modules.setdefault("torchvision.transforms", tv_tf)
    sys.modules.setdefault("torchvision.transforms.functional", tv_tf.functional)
    # open3d (object_point_cloud_map.py:7,186-192): PointCloud.points / Vector3dVector / cluster_dbscan only
    from . import ref_object_map as rom

    class _PointCloud:
        points = None

        def cluster_dbscan(self, eps, min_points):
            return rom.cluster_dbscan(np.asarray(self.points), eps, min_points).tolist()

    if real and _real_or_none("open3d") is not None:
        _backends["open3d"] = "real"
    else:
        _backends["open3d"] = "stand-in: oracle/ref_object_map.py cluster_dbscan"
    o3d = types.ModuleType("open3d")
    _mark(o3d)
    o3d.geometry = types.SimpleNamespace(PointCloud=_PointCloud)
    o3d.utility = types.SimpleNamespace(Vector3dVector=lambda a: np.asarray(a))
    sys.modules.setdefault("open3d", o3d)


def reference_policy():
    """(vlfm.policy.itm_policy, vlfm.policy.base_objectnav_policy) of the real reference.

    Needs two more stand-ins, both for things OUTSIDE the hot path (SURVEY.md section 8, row a24 is control flow only):
    ``hydra.core.config_store.ConfigStore`` (base_objectnav_policy.py:10,387-388 registers a dataclass at import time)
    and ``vlfm.policy.utils.pointnav_policy.WrappedPointNavResNetPolicy`` (needs gym + weights; the generator overrides
    ``_pointnav`` anyway).  habitat_policies.py itself is not importable (habitat, omegaconf, depth_camera_filtering):
    the few lines of HabitatMixin the episode needs are restated in tests/golden/make_golden.py, cited there."""
    install()
    import importlib

    if "hydra.core.config_store" not in sys.modules:
        class ConfigStore:  # noqa: D401 - registration sink
            _inst = None

            @classmethod
            def instance(cls):
                cls._inst = cls._inst or cls()
                return cls._inst

            def store(self, *a, **k):
                return None

        hydra, core, store = (types.ModuleType(n) for n in ("hydra", "hydra.core", "hydra.core.config_store"))
        hydra.__path__, core.__path__ = [], []
        store.ConfigStore = ConfigStore
        hydra.core, core.config_store = core, store
        sys.modules.update({"hydra": hydra, "hydra.core": core, "hydra.core.config_store": store})
    if "vlfm.policy.utils.pointnav_policy" not in sys.modules:
        pn = types.ModuleType("vlfm.policy.utils.pointnav_policy")

        class WrappedPointNavResNetPolicy:
            def __init__(self, *a, **k):
                self.resets = 0

            def reset(self):
                self.resets += 1

        pn.WrappedPointNavResNetPolicy = WrappedPointNavResNetPolicy
        sys.modules["vlfm.policy.utils.pointnav_policy"] = pn
    return importlib.import_module("vlfm.policy.itm_policy"), importlib.import_module("vlfm.policy.base_objectnav_policy")


def reference_object_map():
    """vlfm.mapping.object_point_cloud_map of the real reference (cv2.erode/boundingRect + open3d DBSCAN stand-ins)."""
    install()
    import importlib

    return importlib.import_module("vlfm.mapping.object_point_cloud_map")


def reference_detections():
    """vlfm.vlm.detections of the real reference (needs cv2 + torchvision.ops stand-ins only)."""
    install()
    import importlib

    return importlib.import_module("vlfm.vlm.detections")


def reference_modules():
    """(value_map, obstacle_map, geometry_utils, img_utils) modules of the real reference."""
    install()
    import importlib

    return tuple(importlib.import_module(m) for m in (
        "vlfm.mapping.value_map", "vlfm.mapping.obstacle_map", "vlfm.utils.geometry_utils", "vlfm.utils.img_utils"))

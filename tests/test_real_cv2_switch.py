"""the switch that takes the [ext] stand-ins out.  No real cv2 exists in this container, so a RECORDING module named
``cv2`` (it forwards to oracle/cvport.c's facade and counts calls) plays the real wheel: under VLFM_REAL_CV2=1 the reference's own
value_map.py / obstacle_map.py and the oracle restatements must reach it, and nothing may be planted over it.  Each case runs in its
own process (the switch is read when oracle.cv is imported)."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REFERENCE = os.path.isfile("/root/reference/vlfm/mapping/value_map.py")

FAKE_CV2 = '''
"""A recording cv2: every call is counted in CALLS, the work is done by oracle/cvport.c's facade (STANDIN)."""
__version__ = "4.5.5-recording"
CALLS = {}
from oracle import cv as _facade
RETR_EXTERNAL, RETR_LIST, RETR_CCOMP, RETR_TREE, CHAIN_APPROX_NONE, CHAIN_APPROX_SIMPLE = 0, 1, 2, 3, 1, 2
COLORMAP_INFERNO = COLOR_BGR2RGB = COLOR_GRAY2RGB = COLOR_GRAY2BGR = INTER_AREA = BORDER_CONSTANT = IMREAD_GRAYSCALE = 0
FONT_HERSHEY_SIMPLEX = 0
def _rec(name):
    def f(*a, **k):
        CALLS[name] = CALLS.get(name, 0) + 1
        return _facade.STANDIN[name](*a, **k)
    return f
for _n in _facade.FACADE_NAMES:
    globals()[_n] = _rec(_n)
'''


def run(code: str, tmp_path, real: bool):
    (tmp_path / "cv2.py").write_text(FAKE_CV2)
    env = dict(os.environ, PYTHONPATH=f"{tmp_path}{os.pathsep}{ROOT}")
    env.pop("VLFM_REAL_CV2", None)
    if real:
        env["VLFM_REAL_CV2"] = "1"
    out = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=600,
                         cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


ORACLE_CODE = """
    import json, sys
    import numpy as np
    from oracle import cv
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd.synthetic import SyntheticEnv, MIN_DEPTH, MAX_DEPTH, camera_intrinsics
    fx, fy, fov = camera_intrinsics(640)
    env = SyntheticEnv(3)
    vm = RefValueMap(1, use_max_confidence=False)
    for _ in range(2):
        depth, tf, values = env.observe()
        vm.update_map(values, depth, tf, MIN_DEPTH, MAX_DEPTH, fov)
    m = sys.modules.get("cv2")
    print(json.dumps({"backend": cv.BACKEND, "calls": getattr(m, "CALLS", None), "nonzero": int((vm._map > 0).sum()),
                      "bound": cv.ellipse is getattr(m, "ellipse", None)}))
"""


def test_oracle_facade_is_rebound_to_the_importable_cv2(tmp_path):
    real = run(ORACLE_CODE, tmp_path, real=True)
    assert real["backend"].startswith("real: cv2 4.5.5-recording") and real["bound"]
    assert real["calls"] and real["calls"].get("warpAffine", 0) >= 2 and real["calls"].get("drawContours", 0) >= 2, real
    plain = run(ORACLE_CODE, tmp_path, real=False)       # same machine, switch off: the cv2 on the path is never imported
    assert plain["backend"].startswith("stand-in") and plain["calls"] is None and not plain["bound"]
    assert plain["nonzero"] == real["nonzero"] > 0


REFERENCE_CODE = """
    import json, sys
    import numpy as np
    from oracle import ref_shim
    ref_vm, ref_om, geo, img = ref_shim.reference_modules()
    import cv2
    from vlfm_amd.synthetic import SyntheticEnv, MIN_DEPTH, MAX_DEPTH, camera_intrinsics
    fx, fy, fov = camera_intrinsics(640)
    env = SyntheticEnv(1)
    vm = ref_vm.ValueMap(value_channels=1, use_max_confidence=False)
    om = ref_om.ObstacleMap(min_height=0.61, max_height=0.88, agent_radius=0.18, area_thresh=1.5)
    for _ in range(2):
        depth, tf, values = env.observe()
        om.update_map(depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fx, fy, fov)
        vm.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fov)
    print(json.dumps({"backends": ref_shim.backends(), "standin": bool(getattr(cv2, "__vlfm_standin__", False)),
                      "calls": getattr(cv2, "CALLS", None), "reference_file": ref_vm.__file__,
                      "frontiers": int(len(om.frontiers))}))
"""


@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference absent")
def test_reference_sources_run_on_the_importable_cv2_not_on_the_standin(tmp_path):
    real = run(REFERENCE_CODE, tmp_path, real=True)
    assert real["reference_file"].startswith("/root/reference/")
    assert not real["standin"] and real["backends"]["cv2"].startswith("real 4.5.5-recording")
    for name in ("ellipse", "warpAffine", "dilate", "findContours"):    # value_map.py:260,325-334, obstacle_map.py:117,128,164
        assert real["calls"].get(name, 0) > 0, (name, real["calls"])
    plain = run(REFERENCE_CODE, tmp_path, real=False)
    assert plain["standin"] and plain["calls"] is None and plain["backends"]["cv2"].startswith("stand-in")
    assert plain["frontiers"] == real["frontiers"]


def test_the_switch_fails_loudly_without_a_cv2(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT, VLFM_REAL_CV2="1")
    out = subprocess.run([sys.executable, "-c", "from oracle import cv"], env=env, capture_output=True, text=True, timeout=120,
                         cwd=str(tmp_path))
    assert out.returncode != 0 and "cv2" in out.stderr

"""-m gpu: INTEGRATION.md section B -- the ctypes binding a reference maintainer would write -- is EXECUTED as printed: the
first ```python block after the "## B." heading is extracted, run in an empty namespace (it imports only ctypes / numpy /
torch and dlopens the library) and its ValueMap is driven next to the oracle."""
import os
import re

import numpy as np
import pytest

from vlfm_amd.synthetic import MAX_DEPTH, MIN_DEPTH, SyntheticEnv, camera_intrinsics

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## B. Binding the C ABI yourself"):]
    m = re.search(r"```python\n(.*?)\n```", section, re.S)
    assert m, "INTEGRATION.md section B has no python block"
    return m.group(1)


def test_the_printed_ctypes_stub_runs_and_matches_the_oracle(gpu_device, monkeypatch):
    from oracle.ref_value_map import RefValueMap
    from vlfm_amd import _lib

    src = _stub_source()
    assert "vlfm_amd" not in src.replace("libvlfm_amd.so", "").replace("VLFM_AMD_LIB", "")   # nothing of the Python layer
    monkeypatch.setenv("VLFM_AMD_LIB", _lib.LIB_PATH)
    ns = {}
    exec(compile(src, "INTEGRATION.md#B", "exec"), ns)
    fov = camera_intrinsics(640)[2]
    for use_max in (False, True):
        env = SyntheticEnv(4)
        ours, ref = ns["ValueMap"](1, use_max_confidence=use_max), RefValueMap(1, use_max_confidence=use_max)
        for _ in range(12):
            depth, tf, values = env.observe()
            ours.update_map(values, depth, tf, MIN_DEPTH, MAX_DEPTH, fov)
            ref.update_map(values, depth.copy(), tf, MIN_DEPTH, MAX_DEPTH, fov)
        conf, value = ours._conf[0].cpu().numpy(), ours._value[0].cpu().numpy()
        assert np.array_equal(conf > 0, ref._map > 0)
        assert np.abs(conf - ref._map).max() <= 1e-6 and np.abs(value - ref._value_map).max() <= 1e-6
        wps = np.random.default_rng(3).uniform(-4, 4, size=(10, 2))
        a, av = ours.sort_waypoints(wps, 0.5)
        b, bv = ref.sort_waypoints(wps, 0.5)
        av, bv = np.asarray(av, float), np.asarray(bv, float)
        assert np.abs(av - bv).max() <= 1e-6
        for i in np.flatnonzero((np.asarray(a) != np.asarray(b)).any(axis=1)):   # same order, up to ties below the table's 1e-6
            j = int(np.flatnonzero((np.asarray(b) == np.asarray(a)[i]).all(axis=1))[0])
            assert abs(bv[j] - bv[i]) <= 2e-6
    with pytest.raises(AssertionError, match="outside the image"):
        far = np.eye(4); far[0, 3] = 30.0
        ours.update_map(np.array([0.3]), depth, far, MIN_DEPTH, MAX_DEPTH, fov)

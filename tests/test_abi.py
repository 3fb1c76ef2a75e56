"""The C-ABI library loads on a CPU-only box and exports every symbol include/vlfm_amd.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "vlfm_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vlfm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vlfm_amd import _lib

    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/vlfm_amd.h but not exported"
    assert _lib.lib().vlfm_abi_version() >= 1


def test_struct_layouts_match_header():
    from vlfm_amd import _lib

    assert ctypes.sizeof(_lib.VmPose) == 64
    assert ctypes.sizeof(_lib.IngestParams) == 152
    assert _lib.IngestParams.fx.offset == 112 and _lib.IngestParams.env.offset == 144
    # the NumPy views the host layer fills must agree with the ctypes mirrors field by field
    from vlfm_amd.mapping.value_map import INGEST_DTYPE, VM_POSE_DTYPE

    for dt, ct in ((VM_POSE_DTYPE, _lib.VmPose), (INGEST_DTYPE, _lib.IngestParams)):
        assert dt.itemsize == ctypes.sizeof(ct)
        for name, _ in ct._fields_:
            assert dt.fields[name][1] == getattr(ct, name).offset, name


def test_no_cpu_fallback_and_no_oracle_import_in_product():
    """The product package must never import oracle/ (it is test infrastructure) and must refuse to run without a GPU."""
    import torch
    import pytest

    for dirpath, _, files in os.walk(os.path.join(ROOT, "vlfm_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports oracle"
                assert "from oracle" not in src and "import oracle" not in src, f
    if not torch.cuda.is_available():
        from vlfm_amd.mapping import ValueMap

        with pytest.raises(RuntimeError, match="no CPU fallback"):
            ValueMap(1)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _no_reference_standins_leak():
    """oracle/ref_shim.py plants stand-in packages (cv2, torchvision, ...) in sys.modules to import the reference's own
    source; take them out again after every test so that nothing imported later (transformers probes for torchvision) can
    mistake a stand-in for the real package."""
    yield
    shim = sys.modules.get("oracle.ref_shim")
    if shim is not None:
        shim.uninstall()

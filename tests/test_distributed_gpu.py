"""-m gpu: two ranks over RCCL when the box has two devices (skipped on the 1-GPU lease): the launcher of bench.py, one
process per GPU, the process group on backend "nccl" (= RCCL), the barriers and the two metric all-reduces around REAL steps
of the harness -- so that the driver's 8-GPU scaling run is not the first time RCCL sees nranks > 1."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_over_rccl_when_two_devices_are_visible():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the gpurun lease has one)")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--envs", "8", "--preroll", "5", "--no-small", "--no-cpu-baseline"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["ranks"] == 2 and r["backend"].startswith("nccl")
    assert r["config"]["global_envs"] == 16 and r["value"] > 0 and r["host"]["busy_cores_all_ranks"] > 0

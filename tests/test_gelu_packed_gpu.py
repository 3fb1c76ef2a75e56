"""-m gpu: the packed-f32 GELU of the fc1 epilogue (csrc/gelu_f16.h: gelu_erf2) is the scalar form (gelu_erf) bit for bit, for EVERY
f32 input below 2^127 in both packed positions -- a standalone HIP program (tools/native/gelu_pk_check.hip, ~1 s on the GPU) that
includes the header the GEMM kernels include.  The scalar form's accuracy against f64 is tools/gemm_f16_probe.py's and
tests/test_gemm_f16_gpu.py's business; this test pins the rewrite to it."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_packed_gelu_equals_scalar_gelu_for_every_f32(gpu_device, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "tools", "native", "gelu_pk_check.hip")
    exe = str(tmp_path / "gelu_pk_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "vlfm_amd", "csrc"),
                    src, "-o", exe], check=True, capture_output=True, timeout=300)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert " 0 differing" in run.stdout, run.stdout

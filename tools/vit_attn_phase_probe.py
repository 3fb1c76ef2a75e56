"""Phase breakdown of vit_attention_kernel (last workgroup, wavefront 0).  Diagnostic build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DVLFM_PHASE_TIMING \
          -o scratch/libvlfm_amd_phase.so vlfm_amd/csrc/*.hip vlfm_amd/csrc/host.cpp
    VLFM_LIB_PATH=$PWD/scratch/libvlfm_amd_phase.so python tools/vit_attn_phase_probe.py 128"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vlfm_amd import _lib
if "VLFM_LIB_PATH" not in os.environ:   # build the diagnostic library here: only vit_attention.hip is recompiled
    _lib.build()
    csrc = os.path.join(ROOT, "vlfm_amd", "csrc")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    obj, out = os.path.join(ROOT, "gpurun_out", "vit_attention_phase.o"), os.path.join(ROOT, "gpurun_out", "libvlfm_amd_attn_phase.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                           "-DVLFM_PHASE_TIMING", "-c", os.path.join(csrc, "vit_attention.hip"), "-o", obj])
    objs = [os.path.join(csrc, "build", f) for f in os.listdir(os.path.join(csrc, "build"))
            if f.endswith(".o") and not f.startswith("vit_attention")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj] + objs)
    os.environ["VLFM_LIB_PATH"] = out
    import importlib
    importlib.reload(_lib)
import numpy as np, torch
from vlfm_amd.vlm import ops
dev = torch.device("cuda:0")
names = ["issue K/V loads", "Q loads + LDS writes", "barrier", "QK^T (54 MFMA)", "softmax", "-", "PV (54 MFMA)", "store", "CLS tile", "barrier", "CLS merge"]
for B in (int(a) for a in (sys.argv[1:] or ["128"])):
    qkv = torch.randn(B * 257, 3 * 16 * 88, device=dev, dtype=torch.float16)
    acc = np.zeros(11); n = 0
    for _ in range(12):
        ops.vit_attention(qkv, B, 257, 16, 88, 88 ** -0.5); torch.cuda.synchronize()
        buf = np.zeros(16, np.int64)
        _lib.lib().vlfm_debug_attention_clocks(ctypes.c_void_p(buf.ctypes.data))
        d = np.diff(buf[:12]) * 0.01
        d[5] = 0  # stamp 6 unused: phase 5 -> 7 is PV
        d[6] = (buf[7] - buf[5]) * 0.01
        if _ >= 2: acc += d; n += 1
    print(f"B={B}: " + "  ".join(f"{nm}={a / n:.2f}us" for nm, a in zip(names, acc) if nm != "-") + f"   total={acc.sum() / n:.2f}us")

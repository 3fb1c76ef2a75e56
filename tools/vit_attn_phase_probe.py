"""Phase breakdown of vit_attention_kernel (last workgroup, wavefront 0).  Diagnostic build:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DVLFM_PHASE_TIMING \
          -o scratch/libvlfm_amd_phase.so vlfm_amd/csrc/*.hip vlfm_amd/csrc/host.cpp
    VLFM_LIB_PATH=$PWD/scratch/libvlfm_amd_phase.so python tools/vit_attn_phase_probe.py 128"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vlfm_amd import _lib
from vlfm_amd.vlm import ops
dev = torch.device("cuda:0")
names = ["issue K/V loads", "Q loads + LDS writes", "barrier", "QK^T (54 MFMA)", "softmax", "-", "PV (54 MFMA)", "store", "CLS tile", "barrier", "CLS merge"]
for B in (int(a) for a in (sys.argv[1:] or ["128"])):
    qkv = torch.randn(B * 257, 3 * 16 * 96, device=dev, dtype=torch.float16)
    acc = np.zeros(11); n = 0
    for _ in range(12):
        ops.vit_attention(qkv, B, 257, 16, 96, 88 ** -0.5); torch.cuda.synchronize()
        buf = np.zeros(16, np.int64)
        _lib.lib().vlfm_debug_attention_clocks(ctypes.c_void_p(buf.ctypes.data))
        d = np.diff(buf[:12]) * 0.01
        d[5] = 0  # stamp 6 unused: phase 5 -> 7 is PV
        d[6] = (buf[7] - buf[5]) * 0.01
        if _ >= 2: acc += d; n += 1
    print(f"B={B}: " + "  ".join(f"{nm}={a / n:.2f}us" for nm, a in zip(names, acc) if nm != "-") + f"   total={acc.sum() / n:.2f}us")

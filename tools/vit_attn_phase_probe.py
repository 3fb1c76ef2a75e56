"""Phase stamps of the persistent vit_attention kernel (workgroup 0, wavefronts 0 and 4, every item they walk).  Builds a diagnostic
library (-DVLFM_PHASE_TIMING, only vit_attention.hip recompiled) under gpurun_out/ and prints cycles per phase (shader clock)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vlfm_amd import _lib
if "VLFM_LIB_PATH" not in os.environ:
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
names = ["wait K,Q", "barrier A", "Qread Oout mrg", "odd scores", "QK^T+V dma", "C,softmax+Qdma", "wait V", "barrier B", "odd PV", "PV+K dma", "normalise", "(loop)"]
for B in (int(a) for a in (sys.argv[1:] or ["256"])):
    qkv = torch.randn(B * 257, 3 * 16 * 88, device=dev, dtype=torch.float16)
    for _ in range(3):
        ops.vit_attention(qkv, B, 257, 16, 88, 88 ** -0.5); torch.cuda.synchronize()
    buf = np.zeros((2, 20, 12), np.int64)
    _lib.lib().vlfm_debug_attention_pers_clocks(ctypes.c_void_p(buf.ctypes.data))
    items = min(20, (B + 7) // 8 * 16 // 32) or 1
    for w in range(2):
        print(f"B={B} wavefront {4 * w}: cycles per phase, items 0..{items - 1} (shader clock; 100 cycles ~ 0.045 us)")
        print("      " + " ".join(f"{nm[:15]:>15s}" for nm in names) + "       total")
        for n in range(items):
            s = buf[w, n]
            nxt = buf[w, n + 1, 0] if n + 1 < items else s[11]
            d = list(np.diff(s)) + [nxt - s[11]]
            print(f"  {n:2d}: " + " ".join(f"{int(x):15d}" for x in d) + f"   {int(nxt - s[0]):9d}")
        print(f"   first stamp -> last stamp: {int(buf[w, items - 1, 11] - buf[w, 0, 0])} cycles")

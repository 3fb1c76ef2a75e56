"""What bounds vit_attention_kernel?  Diagnostic builds of csrc/vit_attention.hip with parts of the kernel compiled out
(-DVLFM_ATT_STUB=N, see the kernel file), timed at 256 images next to the product kernel:
    python tools/vit_attn_stub_probe.py build      (here: cross-compiles tools/native/libvlfm_att_stub{1,2,3,4}.so)
    python tools/vit_attn_stub_probe.py [images]   (on the GPU box)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "product", 1: "memory side only (no attention)", 2: "no global loads", 3: "no CLS share", 4: "no stores"}
csrc = os.path.join(ROOT, "vlfm_amd", "csrc")
nat = os.path.join(ROOT, "tools", "native")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    from vlfm_amd import _lib
    _lib.build()
    objs = [os.path.join(csrc, "build", f) for f in os.listdir(os.path.join(csrc, "build")) if f.endswith(".o") and not f.startswith("vit_attention")]
    for n in (1, 2, 3, 4):
        obj = os.path.join(nat, f"vit_attention_stub{n}.o")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-DVLFM_ATT_STUB={n}",
                               "-c", os.path.join(csrc, "vit_attention.hip"), "-o", obj])
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(nat, f"libvlfm_att_stub{n}.so"), obj] + objs)
        os.remove(obj)
    sys.exit(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
if "VLFM_ATT_STUB_CHILD" not in os.environ:
    for n in (0, 1, 2, 3, 4, 0):
        env = dict(os.environ, VLFM_ATT_STUB_CHILD=str(n))
        if n: env["VLFM_LIB_PATH"] = os.path.join(nat, f"libvlfm_att_stub{n}.so")
        subprocess.call([sys.executable, __file__, str(B)], env=env)
    sys.exit(0)
import torch
from vlfm_amd.vlm import ops
n = int(os.environ["VLFM_ATT_STUB_CHILD"])
dev = torch.device("cuda:0")
qkv = torch.randn(B * 257, 3 * 16 * 88, device=dev, dtype=torch.float16)
for _ in range(5): ops.vit_attention(qkv, B, 257, 16, 88, 88 ** -0.5)
ts = []
for _ in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.vit_attention(qkv, B, 257, 16, 88, 88 ** -0.5)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20)
ts.sort()
print(f"{B} images, {NAMES[n]:34s}: {ts[3]*1e6:7.1f} us (min {ts[0]*1e6:.1f})")

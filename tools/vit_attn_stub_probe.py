"""What bounds the persistent vit_attention kernel?  Diagnostic builds of csrc/vit_attention.hip (-DPA_STUB=n) timed next to the product:
1 = no MFMAs (a short sleep per step instead), 2 = 1 + no softmax, 3 = 2 + no odd-query passes, 4 = 3 without the sleeps (memory traffic + barriers), 5 = 4 without the stores."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
B = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 256
if "--child" not in sys.argv:
    from vlfm_amd import _lib
    _lib.build()
    csrc, nat = os.path.join(ROOT, "vlfm_amd", "csrc"), os.path.join(ROOT, "gpurun_out")
    os.makedirs(nat, exist_ok=True)
    objs = [os.path.join(csrc, "build", f) for f in os.listdir(os.path.join(csrc, "build")) if f.endswith(".o") and not f.startswith("vit_attention")]
    extra = [a for a in sys.argv[2:] if a.startswith("-D")]
    for n in [int(a) for a in os.environ.get("PA_STUBS", "0,1,2,3,4,5").split(",")]:
        obj, so = os.path.join(nat, f"vit_attention_stub{n}.o"), os.path.join(nat, f"libvlfm_amd_attn_stub{n}.so")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"-DPA_STUB={n}"] + extra +
                              ["-c", os.path.join(csrc, "vit_attention.hip"), "-o", obj])
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj] + objs)
        env = dict(os.environ, VLFM_LIB_PATH=so)
        subprocess.check_call([sys.executable, __file__, str(B)] + extra + ["--child", str(n)], env=env)
    sys.exit(0)
import torch
from vlfm_amd.vlm import ops
dev = torch.device("cuda:0")
qkv = torch.randn(B * 257, 3 * 16 * 88, device=dev, dtype=torch.float16)
for _ in range(5): ops.vit_attention(qkv, B, 257, 16, 88, 88 ** -0.5)
ts = []
for rep in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ops.vit_attention(qkv, B, 257, 16, 88, 88 ** -0.5)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
print(f"{B} images, {' '.join(a for a in sys.argv if a.startswith('-D'))} PA_STUB={sys.argv[-1]}: {sum(ts) / len(ts):7.1f} us (min {min(ts):.1f})")

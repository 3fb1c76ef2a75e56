"""Which convolutions of the full step's networks (MobileSAM's TinyViT, PointNav's ResNet-18, the YOLOv7-class stand-in) are slow?
Hooks every nn.Conv2d during one real forward, then times each distinct (input, weight, stride, padding, groups, dtype) alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, torch.nn.functional as F
torch.set_num_threads(1)
dev = torch.device("cuda:0")
seen = {}
def hook(name):
    def f(m, inp, out):
        x = inp[0]
        key = (tuple(x.shape), tuple(m.weight.shape), m.stride, m.padding, m.groups, str(x.dtype), x.is_contiguous(memory_format=torch.channels_last))
        seen.setdefault(key, []).append(name)
    return f
def attach(model, tag):
    for n, m in model.named_modules():
        if isinstance(m, nn.Conv2d): m.register_forward_hook(hook(tag + "." + n))
def timed(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
from vlfm_amd.vlm.sam import MobileSAM
from vlfm_amd.pointnav import WrappedPointNavResNetPolicy
from vlfm_amd.vlm.yolov7 import YOLOv7
sam = MobileSAM(device=dev, allow_random_init=True)
pn = WrappedPointNavResNetPolicy(None, device=dev, n_envs=E, discrete_actions=True)
yolo = YOLOv7(device=dev, allow_random_init=True)
for obj, tag in ((sam, "sam"), (pn, "pointnav"), (yolo, "yolo")):
    for attr in dir(obj):
        v = getattr(obj, attr, None)
        if isinstance(v, nn.Module): attach(v, tag + "." + attr)
img = torch.randint(0, 256, (E // 4, 480, 640, 3), dtype=torch.uint8, device=dev)
boxes = torch.tensor([[100., 100., 300., 300.]], device=dev).repeat(E // 4, 1).view(E // 4, 1, 4)
with torch.inference_mode():
    sam.segment_bboxes(img, boxes)
    depth = torch.rand(E, 480, 640, 1, device=dev)
    pn.act_on_depth(depth, torch.rand(E, 2, device=dev), torch.ones(E, 1, dtype=torch.bool, device=dev))
    yolo.predict_batch(torch.randint(0, 256, (E, 480, 640, 3), dtype=torch.uint8, device=dev))
rows = []
with torch.inference_mode():
    for (xs, ws, st, pd, g, dt, cl), names in seen.items():
        dtype = torch.float16 if "16" in dt else torch.float32
        x = torch.randn(xs, device=dev, dtype=dtype); w = torch.randn(ws, device=dev, dtype=dtype)
        if cl: x = x.contiguous(memory_format=torch.channels_last)
        t = timed(lambda: F.conv2d(x, w, None, st, pd, 1, g))
        fl = 2.0 * xs[0] * ws[0] * (xs[2] // st[0]) * (xs[3] // st[1]) * ws[1] * ws[2] * ws[3]
        rows.append((t * len(names), t, len(names), fl / t / 1e6, xs, ws, st, pd, g, dt, names[0]))
rows.sort(reverse=True)
print(f"{'total us':>9} {'each us':>8} n {'TF/s':>6}  input / weight / stride / pad / groups / dtype / first module")
for r in rows[:28]:
    print(f"{r[0]:9.0f} {r[1]:8.0f} {r[2]} {r[3]:6.1f}  {r[4]} {r[5]} {r[6]} {r[7]} {r[8]} {r[9]} {r[10]}")

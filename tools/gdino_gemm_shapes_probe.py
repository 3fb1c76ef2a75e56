"""Which Linear shapes does the fused GroundingDINO forward send to csrc/gemm_f32.hip at B frames, and what does each cost?
Eager forward (no graph), every ops.linear_f32 call bracketed by events.  Also times every F.linear / bmm it still issues.
    python tools/gdino_gemm_shapes_probe.py [B]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from vlfm_amd.vlm import ops
from vlfm_amd.vlm.grounding_dino import GroundingDINO
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
g = GroundingDINO(device=dev, allow_random_init=True, graph=False)
cap = "chair . bed . potted plant . toilet . tv . couch ."
img = torch.randint(0, 256, (B, 480, 640, 3), dtype=torch.uint8, device=dev)
for _ in range(2):
    g.predict_batch(img, [cap])
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
real = ops.linear_f32
def timed(x, weight, bias=None, act=None, residual=None, precision="exact", out=None, owner=""):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    y = real(x, weight, bias, act=act, residual=residual, precision=precision, out=out, owner=owner)
    e.record(); e.synchronize()
    key = (x.numel() // x.shape[-1], weight.shape[0], weight.shape[1], act, residual is not None, precision)
    acc[key][0] += 1; acc[key][1] += s.elapsed_time(e) * 1e3
    return y
ops.linear_f32 = timed
import vlfm_amd.vlm.gdino_fast as gf
gf.ops.linear_f32 = timed
g.predict_batch(img, [cap])
torch.cuda.synchronize()
tot = sum(v[1] for v in acc.values())
print(f"B={B}: {sum(v[0] for v in acc.values())} linear_f32 calls, {tot / 1e3:.2f} ms (event-bracketed, eager)")
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    M, N, K = k[0], k[1], k[2]
    print(f"  M={M:7d} N={N:5d} K={K:5d} act={str(k[3]):5s} res={int(k[4])} {k[5]:6s}: {n:3d} calls, {t / n:7.1f} us each, {t / 1e3:6.2f} ms, {2.0 * M * N * K * n / t / 1e6:7.1f} TFLOP/s")

"""LayerNorm(x + c) HIP kernel vs torch.layer_norm (+ separate add) at the ViT-g stream shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlfm_amd.vlm import ops
dev = torch.device("cuda:0")
for E in (8, 32, 128):
    x = torch.randn(E * 257, 1408, device=dev, dtype=torch.float16)
    c = torch.randn(1408, device=dev); w = torch.ones(1408, device=dev, dtype=torch.float16); b = torch.zeros(1408, device=dev, dtype=torch.float16)
    def t(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    a = t(lambda: ops.layernorm_bias(x, c, w, b, 1e-6)); bb = t(lambda: torch.nn.functional.layer_norm(x, (1408,), w, b, 1e-6))
    gb = 2 * x.numel() * 2 / 1e9
    print(f"E={E}: hip LN(x+c) {a:7.1f} us ({gb/a*1e6/1e3:5.2f} TB/s)   torch LN {bb:7.1f} us ({gb/bb*1e6/1e3:5.2f} TB/s)")

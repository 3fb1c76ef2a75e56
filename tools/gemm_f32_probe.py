"""TFLOP/s of csrc/gemm_f32.hip (exact / split) against torch (hipBLASLt) f32 on GroundingDINO's GEMM shapes at 64 frames.
    python tools/gemm_f32_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlfm_amd.vlm import ops
dev = torch.device("cuda:0")
B = 64
SHAPES = [("swin1 qkv-like 96->96", B * 20286, 96, 96), ("swin1 fc1 96->384", B * 19200, 384, 96), ("swin1 fc2 384->96", B * 19200, 96, 384),
          ("swin2 fc1 192->768", B * 4800, 768, 192), ("swin3 fc1 384->1536", B * 1200, 1536, 384), ("swin3 fc2 1536->384", B * 1200, 384, 1536),
          ("swin4 fc1 768->3072", B * 300, 3072, 768), ("enc ffn1 256->2048", B * 6380, 2048, 256), ("enc ffn2 2048->256", B * 6380, 256, 2048),
          ("enc fusion 256->1024", B * 6380, 1024, 256), ("enc fusion 1024->256", B * 6380, 256, 1024), ("enc 256->256", B * 6380, 256, 256),
          ("dec 256->256", B * 900, 256, 256), ("square 4096", 4096, 4096, 4096)]
def bench(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
tot = {"lib": 0.0, "exact": 0.0, "split": 0.0}
for name, M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    fl = 2.0 * M * N * K
    t_lib = bench(lambda: torch.addmm(b, x, w.t(), out=out))
    t_ex = bench(lambda: ops.linear_f32(x, w, b, precision="exact", out=out))
    t_sp = bench(lambda: ops.linear_f32(x, w, b, precision="split", out=out))
    gb = (M * K + N * K + M * N) * 4 / 1e9
    print(f"{name:26s} M={M:8d} N={N:5d} K={K:5d}  hipBLASLt {fl / t_lib / 1e12:6.1f}  exact {fl / t_ex / 1e12:6.1f}  split {fl / t_sp / 1e12:6.1f} TFLOP/s"
          f"   ({t_lib * 1e3:7.3f} / {t_ex * 1e3:7.3f} / {t_sp * 1e3:7.3f} ms; HBM floor {gb / 6.0 * 1e3:6.3f} ms)", flush=True)
    if "square" not in name:
        tot["lib"] += t_lib; tot["exact"] += t_ex; tot["split"] += t_sp
    del x, w, out
print("sum over the GroundingDINO shapes (one call each):", {k: round(v * 1e3, 2) for k, v in tot.items()}, "ms")

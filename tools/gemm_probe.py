"""ViT-g GEMM shapes at batch E: default hipBLASLt heuristic vs TunableOp, fp16."""
import os, sys, time, torch, torch.nn.functional as F
E = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0"); M = E * 257
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
shapes = {"qkv(pad96)": (1408, 3 * 16 * 96), "proj(pad96)": (16 * 96, 1408), "fc1": (1408, 6144), "fc2": (6144, 1408)}
tot = 0
for name, (k, n) in shapes.items():
    x = torch.randn(M, k, device=dev, dtype=torch.float16); w = torch.randn(n, k, device=dev, dtype=torch.float16) * 0.02; b = torch.randn(n, device=dev, dtype=torch.float16)
    t = bench(lambda: F.linear(x, w, b)); tot += t
    print(f"{name:12s} M={M} K={k} N={n}: {t:8.1f} us  {2*M*k*n/t/1e6:7.0f} TF/s")
print("sum per block %.1f us -> x39 = %.2f ms" % (tot, tot * 39 / 1e3))

import time, torch
dev = torch.device("cuda:0")
M, K, N = 128 * 257, 1408, 1536
x = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.02).half(); b = torch.randn(N, device=dev)
o = torch.addmm(b, x, w.t(), out_dtype=torch.float32)
ref = torch.addmm(o, x, w.t(), alpha=0.5, out_dtype=torch.float32)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("addmm new tensor: %.0f us" % t(lambda: torch.addmm(o, x, w.t(), alpha=0.5, out_dtype=torch.float32)))
try:
    o2 = o.clone()
    torch.addmm(o2, x, w.t(), alpha=0.5, out_dtype=torch.float32, out=o2)
    print("out=self works; max diff vs out-of-place:", float((o2 - ref).abs().max()))
    print("addmm out=self: %.0f us" % t(lambda: torch.addmm(o2, x, w.t(), alpha=0.5, out_dtype=torch.float32, out=o2)))
except Exception as e:
    print("out= variant:", type(e).__name__, str(e)[:300])
print("mm only: %.0f us" % t(lambda: torch.mm(x, w.t(), out_dtype=torch.float32)))

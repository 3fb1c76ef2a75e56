"""Q-Former cross-attention K/V projection: X (f16-exact image tokens) @ W^T (f32 weights).
fp32 GEMM vs three f16 x f16 -> f32 GEMMs on an EXACT three-way split of W (W = W1 + 2^-11 W2 + 2^-22 W3): speed and error
against an f64 reference."""
import time, torch
dev = torch.device("cuda:0")
M, K, N = 128 * 257, 1408, 1536
x16 = torch.randn(M, K, device=dev).half(); w = torch.randn(N, K, device=dev) * 0.02; b = torch.randn(N, device=dev) * 0.1
def split3(w):
    w1 = w.half(); r = (w - w1.float()) * 2048.0
    w2 = r.half(); r = (r - w2.float()) * 2048.0
    w3 = r.half()
    return w1, w2, w3
w1, w2, w3 = split3(w)
rec = w1.double() + w2.double() / 2048 + w3.double() / 2048 ** 2
print("split reconstructs W exactly:", bool((rec == w.double()).all()), " max |W - rec| =", float((rec - w.double()).abs().max()))
def f32_path():
    return torch.nn.functional.linear(x16.float(), w, b)
def split_path():
    o = torch.addmm(b, x16, w3.t(), alpha=2.0 ** -22, out_dtype=torch.float32)
    o = torch.addmm(o, x16, w2.t(), alpha=2.0 ** -11, out_dtype=torch.float32)
    return torch.addmm(o, x16, w1.t(), out_dtype=torch.float32)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
ref = (x16[:4096].double() @ w.double().t() + b.double())
e32 = (f32_path()[:4096].double() - ref).abs(); es = (split_path()[:4096].double() - ref).abs()
print(f"error vs f64: fp32 GEMM max {float(e32.max()):.3e} mean {float(e32.mean()):.3e}   split max {float(es.max()):.3e} mean {float(es.mean()):.3e}")
print(f"fp32 GEMM (incl. x.float()) {t(f32_path):.0f} us   split 3x f16->f32 {t(split_path):.0f} us")
xf = x16.float()
print(f"fp32 GEMM alone {t(lambda: torch.nn.functional.linear(xf, w, b)):.0f} us   one f16->f32 GEMM {t(lambda: torch.mm(x16, w1.t(), out_dtype=torch.float32)):.0f} us")

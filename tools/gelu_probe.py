"""Does the GEMM-epilogue GELU (torch._addmm_activation -> hipBLASLt) implement erf-GELU or the tanh approximation?"""
import time, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
M, K, N = 128 * 257, 1408, 6144
x = torch.randn(M, K, device=dev, dtype=torch.float16); w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.03; b = torch.randn(N, device=dev, dtype=torch.float16)
y_lin = F.linear(x, w, b).float()
erf, tanh = F.gelu(y_lin), F.gelu(y_lin, approximate="tanh")
fused = torch._addmm_activation(b, x, w.t(), use_gelu=True).float()
print("fused vs erf-GELU  max", float((fused - erf).abs().max()), " vs tanh-GELU max", float((fused - tanh).abs().max()), " erf vs tanh max", float((erf - tanh).abs().max()))
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print("linear + gelu: %.1f us   fused: %.1f us   linear only: %.1f us" % (t(lambda: F.gelu(F.linear(x, w, b))), t(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True)), t(lambda: F.linear(x, w, b))))
# decisive check at f32 resolution (the erf / tanh forms differ by up to 4.7e-4)
x32 = torch.randn(4096, 256, device=dev); w32 = torch.randn(512, 256, device=dev) * 0.1; b32 = torch.randn(512, device=dev)
y = F.linear(x32, w32, b32)
f32 = torch._addmm_activation(b32, x32, w32.t(), use_gelu=True)
print("f32: fused vs erf max %.3e   fused vs tanh max %.3e" % (float((f32 - F.gelu(y)).abs().max()), float((f32 - F.gelu(y, approximate="tanh")).abs().max())))
xh, wh, bh = x32.half(), w32.half(), b32.half()
yh = F.linear(xh.float(), wh.float(), bh.float())
fh = torch._addmm_activation(bh, xh, wh.t(), use_gelu=True).float()
print("f16 small: fused vs erf(f32 math) mean %.3e   vs tanh mean %.3e" % (float((fh - F.gelu(yh)).abs().mean()), float((fh - F.gelu(yh, approximate="tanh")).abs().mean())))

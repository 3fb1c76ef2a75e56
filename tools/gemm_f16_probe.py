"""vlfm_gemm_f16_nt (hand-written MFMA GEMM + erf-GELU epilogue) against PyTorch: correctness and time on the ViT-g shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vlfm_amd import _lib
_lib.build()
L = _lib.lib()
dev = torch.device("cuda:0")
def ours(x, w, b, epi):
    M, K = x.shape; N = w.shape[0]
    c = torch.empty((M, N), dtype=torch.float16, device=dev)
    _lib.check(L.vlfm_gemm_f16_nt(x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None, c.data_ptr(), M, N, K, epi,
                                  torch.cuda.current_stream().cuda_stream), "gemm")
    return c
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
torch.manual_seed(0)
# correctness on a small, ragged problem: asymmetric data, M and N tails
for (M, N, K) in [(300, 264, 128), (512, 512, 64), (1000, 776, 1408), (130, 8, 64), (2500, 1408, 192)]:
    x = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * 0.05).half(); b = torch.randn(N, device=dev).half()
    ref = x.float() @ w.float().t() + b.float()
    for epi in (0, 1):
        want = F.gelu(ref) if epi else ref
        got = ours(x, w, b, epi).float()
        err = (got - want).abs().max().item(); scale = want.abs().max().item()
        print(f"M={M} N={N} K={K} epi={epi}: max|err| {err:.3e} (max|ref| {scale:.2f})", "OK" if err <= 2e-3 * max(scale, 1) + 2e-3 else "WRONG")
# time on the real shapes (256 images): fc1 + GELU, qkv, proj, fc2
M = 256 * 257
for name, N, K, gelu in [("fc1+gelu", 6144, 1408, True), ("qkv", 4224, 1408, False), ("proj", 1408, 1408, False), ("fc2", 1408, 6144, False)]:
    x = (torch.randn(M, K, device=dev) * 0.5).half(); w = (torch.randn(N, K, device=dev) * 0.03).half(); b = torch.randn(N, device=dev).half()
    t_lib = timeit(lambda: F.gelu(F.linear(x, w, b)) if gelu else F.linear(x, w, b))
    t_gemm_only = timeit(lambda: F.linear(x, w, b))
    t_ours = timeit(lambda: ours(x, w, b, 1 if gelu else 0))
    fl = 2.0 * M * N * K
    print(f"[{'lock-step' if os.environ.get('VLFM_GEMM_VARIANT') == '1' else 'ping-pong'}] {name:9s} M={M} N={N} K={K}: library {t_lib*1e6:7.1f} us (GEMM alone {t_gemm_only*1e6:7.1f} us = {fl/t_gemm_only/1e15:.2f} PF)  ours {t_ours*1e6:7.1f} us = {fl/t_ours/1e15:.2f} PF")
    got = ours(x, w, b, 1 if gelu else 0).float(); want = F.linear(x, w, b).float(); want = F.gelu(want) if gelu else want
    print("           max|ours - library| =", (got - want).abs().max().item())


# crossover against the library at small batches (fc1 + GELU)
for imgs in (1, 8, 16, 32, 64, 128):
    M = imgs * 257
    x = (torch.randn(M, 1408, device=dev) * 0.5).half(); w = (torch.randn(6144, 1408, device=dev) * 0.03).half(); b = torch.randn(6144, device=dev).half()
    t_lib = timeit(lambda: F.gelu(F.linear(x, w, b)), 30); t_ours = timeit(lambda: ours(x, w, b, 1), 30)
    print(f"fc1+gelu at {imgs:3d} images: library {t_lib*1e6:7.1f} us  ours {t_ours*1e6:7.1f} us")

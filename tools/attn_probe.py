"""SDPA variants for the ViT-g attention shape (B=64, H=16, S=257, D=88), fp16."""
import torch, time, torch.nn.functional as F
dev = torch.device("cuda:0")
import sys
B, H, S, D = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 16, 257, 88
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
qkv = torch.randn(B, S, 3, H, D, device=dev, dtype=torch.float16).permute(2, 0, 3, 1, 4)
q, k, v = qkv[0], qkv[1], qkv[2]
print("default strided   %.1f us" % bench(lambda: F.scaled_dot_product_attention(q, k, v)))
qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
print("default contig    %.1f us" % bench(lambda: F.scaled_dot_product_attention(qc, kc, vc)))
for Dp in (96, 128):
    qp, kp, vp = [F.pad(t, (0, Dp - D)).contiguous() for t in (qc, kc, vc)]
    print(f"padded D={Dp}      %.1f us" % bench(lambda: F.scaled_dot_product_attention(qp, kp, vp, scale=D ** -0.5)))
from torch.nn.attention import sdpa_kernel, SDPBackend
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH):
    try:
        with sdpa_kernel(be):
            print(f"{be.name:22s} %.1f us" % bench(lambda: F.scaled_dot_product_attention(qc, kc, vc)))
    except Exception as e:
        print(be.name, "ERR", str(e)[:100])
def manual():
    s = torch.matmul(qc, kc.transpose(-1, -2)) * (D ** -0.5)
    return torch.matmul(torch.softmax(s, dim=-1), vc)
print("bmm+softmax+bmm   %.1f us" % bench(manual))
try:
    torch.backends.cuda.preferred_rocm_fa_library("ck")
    print("ck fa             %.1f us" % bench(lambda: F.scaled_dot_product_attention(qc, kc, vc)))
except Exception as e:
    print("ck ERR", str(e)[:200])
# GEMM rate at the ViT shapes for reference
x = torch.randn(B * S, 1408, device=dev, dtype=torch.float16); w1 = torch.randn(6144, 1408, device=dev, dtype=torch.float16); b1 = torch.randn(6144, device=dev, dtype=torch.float16)
t = bench(lambda: F.linear(x, w1, b1)); print("fc1 GEMM %.1f us  %.0f TF/s" % (t, 2 * B * S * 1408 * 6144 / t / 1e6))
xb = x.to(torch.bfloat16); wb = w1.to(torch.bfloat16); bb = b1.to(torch.bfloat16)
t = bench(lambda: F.linear(xb, wb, bb)); print("fc1 GEMM bf16 %.1f us  %.0f TF/s" % (t, 2 * B * S * 1408 * 6144 / t / 1e6))

"""vlfm_vit_attention_f16 vs library SDPA at the ViT-g shape (B images, 16 heads of 88, 257 tokens)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from vlfm_amd.vlm import ops
dev = torch.device("cuda:0")
for B, D in [(int(a), d) for a in (sys.argv[1:] or ["128"]) for d in (88,)]:
    S, H = 257, 16
    qkv = torch.randn(B * S, 3 * H * D, device=dev, dtype=torch.float16)
    def t(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    def lib():
        q = qkv.view(B, S, 3, H, D).permute(2, 0, 3, 1, 4)
        return F.scaled_dot_product_attention(q[0], q[1], q[2], scale=88 ** -0.5).transpose(1, 2).reshape(B * S, H * D)
    a, b = t(lambda: ops.vit_attention(qkv, B, S, H, D, 88 ** -0.5)), t(lib)
    fl = 4 * B * H * S * S * 88
    print(f"B={B} D={D}: hip {a:7.1f} us ({fl/a/1e6:5.0f} TF/s)   library sdpa + transpose {b:7.1f} us   max diff {float((ops.vit_attention(qkv, B, S, H, D, 88 ** -0.5).float() - lib().float()).abs().max()):.4f}")

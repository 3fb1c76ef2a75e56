"""TinyViT's window attention shapes (MobileSAM at 32 frames) through the library's scaled_dot_product_attention with the additive
bias, per stage: where the encoder's `attn_fwd` time goes.  Usage: python tools/sam_attn_probe.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, side, ws, heads, blocks in (("stage 1", 128, 7, 4, 2), ("stage 2", 64, 14, 5, 6), ("stage 3", 64, 7, 10, 2)):
    nw = frames * ((side + ws - 1) // ws) ** 2
    n = ws * ws
    q, k, v = (torch.randn(nw, heads, n, 32, device=dev) for _ in range(3))
    bias = torch.randn(1, heads, n, n, device=dev)
    t = timed(lambda: F.scaled_dot_product_attention(q, k, v, attn_mask=bias))
    flop = 4.0 * nw * heads * n * n * 32
    byt = 4.0 * nw * heads * n * 32 * 4
    print(f"{name}: {nw} windows x {heads} heads x {n} tokens: {t:6.3f} ms per block x {blocks} blocks = {t * blocks:6.2f} ms "
          f"({flop / t / 1e9:6.1f} TFLOP/s, {byt / t / 1e9:5.2f} TB/s of q/k/v/o)")

from vlfm_amd.vlm import ops  # noqa: E402

for name, side, ws, heads, blocks in (("stage 1", 128, 7, 4, 2), ("stage 2", 64, 14, 5, 6), ("stage 3", 64, 7, 10, 2)):
    nw = frames * ((side + ws - 1) // ws) ** 2
    n = ws * ws
    qkv = torch.randn(nw, n, heads * 96, device=dev)
    bias_t = torch.randn(heads, n, n, device=dev)
    t = timed(lambda: ops.window_attention(qkv, bias_t, heads, 32 ** -0.5))
    print(f"{name}: csrc/sam_ops.hip window_attention_f32_kernel {t:6.3f} ms per block x {blocks} blocks = {t * blocks:6.2f} ms")

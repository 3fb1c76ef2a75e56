"""vlfm_vit_attention_f16 against softmax(q k^T / sqrt(88)) v in fp32 at several batch sizes (incl. workgroups that walk 2+ items)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlfm_amd.vlm import ops
dev = torch.device("cuda:0")
S, H, D = 257, 16, 88
for B in [int(a) for a in (sys.argv[1:] or ["3", "19", "40"])]:
    g = torch.Generator().manual_seed(B)
    qkv = torch.randn(B, S, 3, H, D, generator=g) * 1.5
    qkv[0, :, 0, 0] *= 4.0
    qkv[B - 1, :, 0, 5] *= 3.0
    half = qkv.half()
    q, k, v = [half[:, :, i].float().permute(0, 2, 1, 3) for i in range(3)]
    want = (torch.softmax(q @ k.transpose(-1, -2) * D ** -0.5, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * S, H * D)
    x = half.to(dev).reshape(B * S, 3 * H * D).contiguous()
    worst = 0.0
    for rep in range(3):
        got = ops.vit_attention(x, B, S, H, D, D ** -0.5).float().cpu()
        err = (got - want).abs()
        worst = max(worst, float(err.max()))
        bad = (err > 6e-3).nonzero()
        if len(bad):
            r, c = int(bad[0][0]), int(bad[0][1])
            print(f"B={B} rep {rep}: {len(bad)} bad, first at image {r // S} token {r % S} head {c // D} ch {c % D}: got {got[r, c]:.4f} want {want[r, c]:.4f}")
            tok = (bad[:, 0] % S); print("   bad tokens hist (0 = CLS):", torch.bincount(tok, minlength=S)[:8].tolist(), "heads:", torch.bincount(bad[:, 1] // D, minlength=H).tolist())
    print(f"B={B}: max err {worst:.5f}  nan {bool(torch.isnan(got).any())}")

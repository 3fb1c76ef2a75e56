"""vlfm_gemm_f16_nt (three epilogues) and vlfm_gemm_f32_nt (exact / split) on RANDOM shapes against f32 / f64 references."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from vlfm_amd.vlm import ops
dev = torch.device("cuda:0")
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b):
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    M = int(rng.choice([1, 7, 64, 255, 256, 257, 300, 511, 513, 1000, 2056, 4112, 5000]))
    N = int(rng.choice([8, 16, 64, 128, 136, 248, 256, 264, 512, 1408, 1416]))
    K = int(rng.choice([64, 128, 192, 576, 1408]))
    x = (torch.randn(M, K, generator=g) * 0.7).half().to(dev); w = (torch.randn(N, K, generator=g) * 0.05).half().to(dev)
    bias = (torch.randn(N, generator=g) * 0.2).half().to(dev); c0 = torch.randn(M, N, generator=g).half().to(dev)
    ref = x.float() @ w.float().t() + bias.float()
    try:
        for epi in ("bias", "bias_gelu", "accumulate"):
            out = c0.clone() if epi == "accumulate" else None
            got = ops.linear_f16(x, w, bias, epi, out=out).float()
            want = F.gelu(ref) if epi == "bias_gelu" else (ref + c0.float() if epi == "accumulate" else ref)
            err = (got - want).abs(); tol = 2e-3 * want.abs().clamp(min=1.0)
            if not bool((err <= tol).all()):
                bad += 1; print("f16 seed", seed, (M, N, K), epi, "max err", float(err.max())); break
            again = ops.linear_f16(x, w, bias, epi, out=c0.clone() if epi == "accumulate" else None).float()
            if not torch.equal(again, got):
                bad += 1; print("f16 seed", seed, (M, N, K), epi, "NOT bitwise repeatable"); break
        # f32 family
        M2 = int(rng.choice([64, 65, 130, 515, 900, 2049])); N2 = int(rng.choice([32, 36, 96, 132, 256, 640, 768, 1152])); K2 = int(rng.choice([32, 64, 96, 256, 768]))
        x32 = torch.randn(M2, K2, generator=g).to(dev); w32 = (torch.randn(N2, K2, generator=g) * 0.05).to(dev); b32 = torch.randn(N2, generator=g).to(dev)
        r64 = x32.double() @ w32.double().t() + b32.double(); scale = x32.double().abs() @ w32.double().abs().t() + b32.double().abs()
        for prec in ("exact", "split"):
            got = ops.linear_f32(x32, w32, b32, precision=prec, owner="stress")
            e = float(((got.double() - r64).abs() / scale).max())
            if e > 3e-6:
                bad += 1; print("f32 seed", seed, (M2, N2, K2), prec, "rel err", e)
    except Exception as e:
        bad += 1; print("seed", seed, (M, N, K), "RAISED", type(e).__name__, str(e)[:200])
print(f"GEMM random shapes, seeds {a}..{b - 1}: {bad} failed")

"""-m gpu: csrc/gemm_f16.hip (the ViT-g GEMMs of BLIP2ITM.cosine, /root/reference/vlfm/vlm/blip2itm.py:37-54 -> LAVIS eva_vit [ext])
against an f64 reference: every epilogue (bias / bias + erf-GELU / accumulate into the residual stream), the persistent 8-phase kernel,
ragged M and N, 1-, 2-, 3-tile and long K (the 8-phase kernel's prologue and tail wait counts), the half-empty last n-tile of
N = 1408 / 4224 -- and a race screen: the 8-phase kernel keeps LDS-DMA loads in flight across barriers, so a wrong wait count is a
RARE wrong tile; the same launch is repeated and compared bitwise.  Tolerance: f16 output rounding (2^-11 relative) + f32
accumulation error of a K-long dot product: |err| <= 2e-3 * max(1, max|ref|)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(300, 264, 128), (512, 512, 64), (256, 256, 192), (1000, 776, 1408), (130, 8, 64), (2500, 1408, 192), (777, 4224, 320),
          (1028, 1408, 6144), (257, 6144, 1408),
          # more tiles than CUs (the persistent kernel walks 2 tiles per workgroup), ragged in m and n, 2 K-tiles and 1
          (7000, 2568, 128), (5000, 4224, 64),
          # fc1 at 32 images: 792 tiles = 3 rounds of 256 + 24, the leftover tiles run as 48 n-half items (both wavefront groups as the
          # only active one)
          (8224, 6144, 128)]


def _run(ops, x, w, b, epi, variant=None, out=None):
    return ops.linear_f16(x, w, b, epi, out=out)      # (one kernel since round 6: the persistent 8-phase schedule)


@pytest.mark.parametrize("variant", [7])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_f16_epilogues_against_f64(gpu_device, shape, variant):
    from vlfm_amd.vlm import ops

    M, N, K = shape
    g = torch.Generator().manual_seed(M * 5 + N * 3 + K + variant)
    x = (torch.randn(M, K, generator=g) * 0.5).half().to(gpu_device)
    w = (torch.randn(N, K, generator=g) * 0.05).half().to(gpu_device)          # asymmetric operands: a transpose cannot hide
    b = torch.randn(N, generator=g).half().to(gpu_device)
    r0 = torch.randn(M, N, generator=g).half().to(gpu_device)
    ref = x.double() @ w.double().t() + b.double()
    for epi in ("bias", "bias_gelu", "accumulate"):
        if epi == "accumulate" and variant < 2:
            continue
        want = F.gelu(ref) if epi == "bias_gelu" else ref + r0.double() if epi == "accumulate" else ref
        out = r0.clone() if epi == "accumulate" else None
        got = _run(ops, x, w, b, epi, variant, out=out).double()
        err = float((got - want).abs().max())
        assert err <= 2e-3 * max(1.0, float(want.abs().max())), (epi, err)
    # no bias, accumulate: what the ViT block's projection / fc2 call
    if variant >= 2:
        got = _run(ops, x, w, None, "accumulate", variant, out=r0.clone()).double()
        want = x.double() @ w.double().t() + r0.double()
        assert float((got - want).abs().max()) <= 2e-3 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("variant", [7])
def test_gemm_f16_8phase_is_deterministic_under_repetition(gpu_device, variant):
    """Race screen: 30 repeats of three multi-wave-of-workgroups problems, bitwise equal to the first run, which is checked against
    an f32 product."""
    from vlfm_amd.vlm import ops

    for (M, N, K) in [(4096, 4224, 1408), (2048, 1408, 6144), (8192, 6144, 1408), (8224, 6144, 1408)]:
        g = torch.Generator().manual_seed(M + N + K)
        x = torch.randn(M, K, generator=g).half().to(gpu_device)
        w = (torch.randn(N, K, generator=g) * 0.05).half().to(gpu_device)
        b = torch.randn(N, generator=g).half().to(gpu_device)
        first = _run(ops, x, w, b, "bias", variant).clone()
        ref = x.float() @ w.float().t() + b.float()
        assert float((first.float() - ref).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max()))
        for _ in range(30):
            again = _run(ops, x, w, b, "bias", variant)
            assert bool((again == first).all())


def test_gemm_f16_rejects_what_it_cannot_do(gpu_device):
    from vlfm_amd.vlm import ops

    x = torch.zeros(64, 96, dtype=torch.float16, device=gpu_device)       # K % 64 != 0
    w = torch.zeros(64, 96, dtype=torch.float16, device=gpu_device)
    assert not ops.linear_f16_supported(x, w)
    with pytest.raises(Exception):
        ops.linear_f16(x, w, None, "bias")
    x = torch.zeros(64, 128, dtype=torch.float16, device=gpu_device)
    w = torch.zeros(64, 128, dtype=torch.float16, device=gpu_device)
    with pytest.raises(AssertionError):
        ops.linear_f16(x, w, None, "accumulate")                          # nothing to accumulate into

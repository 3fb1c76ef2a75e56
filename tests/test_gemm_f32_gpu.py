"""-m gpu: csrc/gemm_f32.hip against an f64 reference on GroundingDINO's Linear shapes (vlfm/vlm/grounding_dino.py:38-74 runs the
network in fp32).  Two forms:
  exact  v_mfma_f32_32x32x2_f32: must be as close to f64 as torch's own f32 GEMM (same rounding class: one rounding per product)
  split  f16 hi/lo operands, three f16 MFMAs: must be F32-GRADE -- its error against f64 at most 2x the exact form's and its
         difference from the f32 result <= 1e-6 of the result's scale ("no split-precision GEMM unless a GPU test shows
         <= 1e-6 relative to f32"); and it must raise the overflow flag instead of returning infinities."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (M, N, K): Swin stage 1 qkv / mlp, stage 3 mlp, encoder FFN, fusion projection, decoder, ragged tails
SHAPES = [(4096, 96, 96), (4096, 384, 96), (2500, 96, 384), (1200, 1536, 384), (3000, 2048, 256), (3000, 256, 2048),
          (6380, 1024, 256), (900, 256, 256), (777, 132, 64), (130, 36, 32),
          # the split form's 256-wide tile (N % 256 == 0 or N >= 640): full tiles, an n tail inside a wide tile, an m tail
          (1000, 768, 192), (515, 2304, 96), (2049, 640, 64), (300, 1152, 384)]


def _err(got, ref64, scale):
    return float(((got.double() - ref64).abs() / scale).max())


@pytest.mark.parametrize("shape", SHAPES)
def test_f32_gemm_forms_against_f64(gpu_device, shape):
    from vlfm_amd.vlm import ops

    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    x = (torch.randn(M, K, generator=g) * torch.logspace(-2, 2, K)[None, :]).to(gpu_device)      # 4 decades of column scales
    w = (torch.randn(N, K, generator=g) * 0.05).to(gpu_device)
    b = torch.randn(N, generator=g).to(gpu_device)
    r = torch.randn(M, N, generator=g).to(gpu_device)
    ref64 = x.double() @ w.double().t() + b.double()
    scale = (x.double().abs() @ w.double().abs().t() + b.double().abs())           # sum |a b|: the natural error scale of a dot product
    lib32 = torch.addmm(b, x, w.t())
    e_lib = _err(lib32, ref64, scale)
    exact = ops.linear_f32(x, w, b, precision="exact")
    e_exact = _err(exact, ref64, scale)
    ops.gemm_f32_overflow_flag(gpu_device).zero_()
    split = ops.linear_f32(x, w, b, precision="split")
    e_split = _err(split, ref64, scale)
    torch.cuda.synchronize()
    assert int(ops.gemm_f32_overflow_flag(gpu_device).item()) == 0
    assert e_exact <= max(2 * e_lib, 2e-7), (shape, e_exact, e_lib)
    assert e_split <= max(2 * e_exact, 3e-7), (shape, e_split, e_exact)
    rel = float((split - exact).abs().max() / exact.abs().max())
    assert rel <= 1e-6, (shape, rel)
    # ... and PER ELEMENT (a bound relative to the largest output says nothing about the small ones): every
    # output within 2e-6 of ITS OWN dot product's scale sum_k |a_k b_k| + |bias| of the exact-f32 result
    per_elem = float(((split.double() - exact.double()).abs() / scale).max())
    assert per_elem <= 2e-6, (shape, per_elem)
    # epilogues: ReLU, exact GELU, residual
    for act, fn in (("relu", torch.relu), ("gelu", torch.nn.functional.gelu)):
        for prec in ("exact", "split"):
            got = ops.linear_f32(x, w, b, act=act, residual=r, precision=prec)
            want = fn(ref64).float() + r
            assert float((got - want).abs().max()) <= 2e-6 * float(scale.max()) + 2e-6, (shape, act, prec)
    # in-place residual (the Linear writes over the residual stream)
    acc = r.clone()
    ops.linear_f32(x, w, None, residual=acc, out=acc, precision="exact")
    assert float((acc - (ref64 - b.double() + r.double()).float()).abs().max()) <= 2e-6 * float(scale.max()) + 2e-6


def test_exact_form_is_a_k_ordered_fma_chain(gpu_device):
    """v_mfma_f32_32x32x2_f32 = fma(a_k, b_k, acc) in k order; the kernel's k permutation inside a 32-float tile is fixed, so
    the result is deterministic and equals a host fma chain in that order (checked on a small case in f64-emulated f32)."""
    import numpy as np

    from vlfm_amd.vlm import ops

    M, N, K = 64, 32, 64
    g = torch.Generator().manual_seed(5)
    x, w = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    got = ops.linear_f32(x.to(gpu_device), w.to(gpu_device), precision="exact").cpu().numpy()
    again = ops.linear_f32(x.to(gpu_device), w.to(gpu_device), precision="exact").cpu().numpy()
    assert np.array_equal(got, again)
    # order inside a K-tile of 32: for kk in 0..3, q in 0..3: k = 8 kk + q (lanes 0-31) then 8 kk + 4 + q (lanes 32-63)
    order = [t * 32 + 8 * kk + half * 4 + q for t in range(K // 32) for kk in range(4) for q in range(4) for half in (0, 1)]
    xn, wn = x.numpy(), w.numpy()
    acc = np.zeros((M, N), np.float32)
    for k in order:
        acc = (xn[:, k:k + 1].astype(np.float64) * wn[None, :, k].astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got, acc), float(np.abs(got - acc).max())


def test_split_form_flags_operands_outside_f16_range(gpu_device):
    from vlfm_amd.vlm import ops

    x = torch.randn(256, 64, device=gpu_device)
    w = torch.randn(64, 64, device=gpu_device)
    flag = ops.gemm_f32_overflow_flag(gpu_device)
    flag.zero_()
    ops.linear_f32(x, w, precision="split")
    assert int(flag.item()) == 0
    x[17, 3] = 7.0e4
    ops.linear_f32(x, w, precision="split")
    assert int(flag.item()) == 1          # sticky until the caller clears it (and repeats the call in the exact form)
    flag.zero_()
    tiny = torch.full((256, 64), 3.0e-6, device=gpu_device)      # f16-subnormal hi parts: the remainder carries the value
    got = ops.linear_f32(tiny, w, precision="split")
    want = tiny.double() @ w.double().t()
    assert float((got.double() - want).abs().max()) <= 1e-6 * float((tiny.double().abs() @ w.double().abs().t()).max())


def test_overflow_flag_is_one_tensor_for_every_spelling_of_the_device(gpu_device):
    from vlfm_amd.vlm import ops

    """ADVICE r4: the kernels are handed the flag of ``x.device`` ("cuda:0"), a network built with ``device="cuda"`` (the
    reference's spelling) must read the SAME tensor, or an overflow is never seen and void results are used."""
    a = ops.gemm_f32_overflow_flag("cuda", "spelling")
    b = ops.gemm_f32_overflow_flag(torch.device("cuda:0"), "spelling")
    c = ops.gemm_f32_overflow_flag(torch.device("cuda", torch.cuda.current_device()), "spelling")
    assert a is b and b is c and a.device.index == 0
    assert ops.gemm_f32_overflow_flag("cuda", "other") is not a


def test_a_weight_outside_f16_range_keeps_raising_the_flag(gpu_device):
    from vlfm_amd.vlm import ops

    """ADVICE r4: the range check of a weight runs when its hi/lo planes are built; the cached planes must not become silent after
    the owner's flag was cleared once."""
    w = torch.randn(64, 64, device=gpu_device)
    w[3, 5] = 1.0e6
    x = torch.randn(128, 64, device=gpu_device)
    flag = ops.gemm_f32_overflow_flag(gpu_device, "badweight")
    flag.zero_()
    with torch.inference_mode():
        ops.linear_f32(x, w, precision="split", owner="badweight")
        assert int(flag.item()) != 0
        flag.zero_()
        assert ops.split_weights_bad(gpu_device, "badweight")
        ops.linear_f32(x, w, precision="split", owner="badweight")      # cache hit: the verdict travels with the planes
        assert int(flag.item()) != 0
    flag.zero_()

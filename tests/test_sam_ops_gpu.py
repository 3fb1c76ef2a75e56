"""csrc/sam_ops.hip -- the NHWC-rows kernels of TinyViT's windowed-attention blocks (MobileSAM's image encoder behind
vlfm/vlm/sam.py:54; TinyViTBlock of mobile_sam/modeling/tiny_vit_sam.py [ext]) against the framework formulation the block used
before (which tests/test_sam_cpu.py pins to the oracle), in f32: |err| <= 2e-5 * max(1, |ref|) per kernel (different summation
orders of a 128-320-term mean / variance and of the 9 filter taps), and the whole encoder on the rows path against the NCHW path
and against the CPU (<= 2e-3 relative to the output's magnitude, the bound the other detector networks use)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def close(a, b, tol=2e-5):
    return bool(((a - b).abs() <= tol * b.abs().clamp(min=1.0)).all())


@pytest.mark.parametrize("B,H,W,C,ws", [(2, 64, 64, 160, 14), (1, 128, 128, 128, 7), (2, 64, 64, 320, 7), (1, 9, 13, 8, 4),
                                        (1, 14, 14, 516, 7)])
def test_layernorm_rows_and_window_partition(gpu_device, B, H, W, C, ws):
    from vlfm_amd.vlm import ops

    g = torch.Generator().manual_seed(C + ws)
    x = (torch.randn(B, H, W, C, generator=g) * 1.7 + 0.3).to(gpu_device)
    gamma, beta = torch.randn(C, generator=g).to(gpu_device), torch.randn(C, generator=g).to(gpu_device)
    ref = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    assert close(ops.layernorm_rows(x, gamma, beta, 1e-5), ref)
    # pad -> partition -> norm, exactly as the block spells it (the padding is normalised too: LayerNorm(0) = beta)
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    t = F.pad(x, (0, 0, 0, pw, 0, ph))
    hp, wp = H + ph, W + pw
    t = t.view(B, hp // ws, ws, wp // ws, ws, C).transpose(2, 3).reshape(-1, ws * ws, C)
    got = ops.layernorm_rows(x, gamma, beta, 1e-5, ws)
    assert got.shape == t.shape and close(got, F.layer_norm(t, (C,), gamma, beta, 1e-5))


@pytest.mark.parametrize("B,H,W,C,ws", [(2, 64, 64, 160, 14), (1, 128, 128, 128, 7), (1, 9, 13, 8, 4)])
def test_window_reverse_add(gpu_device, B, H, W, C, ws):
    from vlfm_amd.vlm import ops

    g = torch.Generator().manual_seed(C * 3 + ws)
    x = torch.randn(B, H, W, C, generator=g).to(gpu_device)
    hp, wp = (H + ws - 1) // ws * ws, (W + ws - 1) // ws * ws
    a = torch.randn(B * (hp // ws) * (wp // ws), ws * ws, C, generator=g).to(gpu_device)
    ref = x + a.view(B, hp // ws, wp // ws, ws, ws, C).transpose(2, 3).reshape(B, hp, wp, C)[:, :H, :W]
    got = ops.window_reverse_add_(x.clone(), a, ws)
    assert torch.equal(got, ref)                 # one f32 addition per element: bit-identical


@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 160), (1, 128, 128, 128), (2, 5, 7, 12), (1, 1, 1, 4)])
def test_depthwise_conv_on_nhwc_rows(gpu_device, B, H, W, C):
    from vlfm_amd.vlm import ops

    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, H, W, C, generator=g).to(gpu_device)
    w = (torch.randn(C, 1, 3, 3, generator=g) / 3).to(gpu_device)
    b = torch.randn(C, generator=g).to(gpu_device)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=1, groups=C).permute(0, 2, 3, 1)
    got = ops.depthwise_conv3x3_nhwc(x, w.reshape(C, 9).t().contiguous(), b)
    assert close(got, ref)
    assert close(ops.depthwise_conv3x3_nhwc(x, w.reshape(C, 9).t().contiguous(), None), ref - b)


def test_tinyvit_encoder_rows_path_matches_the_nchw_path_and_the_cpu(gpu_device):
    from vlfm_amd.vlm import det_ops
    from vlfm_amd.vlm.sam import TinyViT, _TinyViTBlock

    torch.manual_seed(5)
    enc = TinyViT().eval()
    with torch.no_grad():
        for m in enc.modules():          # non-trivial BatchNorm statistics and attention biases, so that folding and the bias matter
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
        for m in enc.modules():
            if hasattr(m, "attention_biases"):
                m.attention_biases.normal_(0, 0.5)
    x = torch.randn(1, 3, 512, 512)      # 1/2 of SAM's side: 32 x 32 maps in the last stages (padded to 35 / 42 by the windows)
    with torch.no_grad():
        cpu = enc(x)[0]
    enc.to(gpu_device)
    det_ops.fold_batchnorm_(enc)
    xg = x.to(gpu_device)
    blocks = [m for m in enc.modules() if isinstance(m, _TinyViTBlock)]
    assert len(blocks) == 10
    with torch.inference_mode():
        assert all(b.rows_path(xg.new_zeros(1, 8, 8, b.local_conv.c.in_channels)) for b in blocks)
        rows = enc(xg)[0].float().cpu()
        for b in blocks:                 # force the NCHW formulation
            b.rows_path = lambda t: False
        nchw = enc(xg)[0].float().cpu()
    scale = float(cpu.abs().max())
    assert scale > 0.1
    assert float((rows - nchw).abs().max()) <= 2e-4 * scale
    assert float((rows - cpu).abs().max()) <= 2e-3 * scale


@pytest.mark.parametrize("nw,n,heads", [(37, 49, 4), (11, 196, 5), (5, 49, 10), (3, 1, 2), (2, 256, 1), (4, 70, 3)])
def test_window_attention(gpu_device, nw, n, heads):
    """vlfm_window_attention_f32 against the library's scaled_dot_product_attention with the additive bias as attn_mask (what the
    block called before), f32: |err| <= 2e-5 * max(1, |ref|) (online softmax over chunks of four keys vs the library's tiling)."""
    from vlfm_amd.vlm import ops

    g = torch.Generator().manual_seed(nw * 100 + n + heads)
    qkv = (torch.randn(nw, n, heads * 96, generator=g) * 1.3).to(gpu_device)
    bias = (torch.randn(heads, n, n, generator=g) * 2.0).to(gpu_device)          # NOT symmetric: the transposition matters
    q, k, v = qkv.view(nw, n, heads, 96).split(32, dim=3)
    ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=bias.unsqueeze(0))
    ref = ref.transpose(1, 2).reshape(nw, n, heads * 32)
    got = ops.window_attention(qkv, bias.transpose(1, 2).contiguous(), heads, 32 ** -0.5)
    assert got.shape == ref.shape and close(got, ref)

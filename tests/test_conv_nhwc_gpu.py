"""vlfm_conv_nhwc_f16 (csrc/conv_nhwc.hip) against torch's convolution in f32 on the same f16 operands: the 1x1 / 3x3 layers of
the yolov7-e6e graph the reference runs in fp16 (vlfm/vlm/yolov7.py:35-48,89) as implicit GEMMs with bias + SiLU fused.
Tolerance: the result is rounded to f16 once (relative 2^-11) on top of an f32 accumulation whose order differs from the
reference's -- |err| <= 2e-3 * max(1, |ref|) per element is written below."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, Cin, Cout, k, stride, H, W): shapes of the e6e graph at 448x640 (tools: the inventory in DESIGN.md section 6d) + edge cases
CASES = [
    (2, 64, 64, 3, 1, 112, 160),     # the heaviest family: 64 -> 64 at stride 4
    (2, 128, 128, 3, 1, 56, 80),
    (1, 256, 256, 3, 1, 28, 40),
    (3, 384, 384, 3, 1, 14, 20),     # three 128-channel tiles
    (2, 512, 512, 3, 1, 7, 10),      # M = 140: one ragged pixel tile
    (2, 192, 192, 3, 1, 14, 20),     # three 64-channel tiles
    (2, 320, 640, 3, 1, 28, 40),
    (2, 640, 256, 1, 1, 28, 40),     # 1x1
    (2, 1280, 320, 1, 1, 28, 40),
    (1, 2560, 1280, 1, 1, 7, 10),
    (2, 320, 320, 3, 2, 56, 80),     # DownC's stride-2 3x3
    (2, 64, 72, 3, 1, 9, 11),        # Cout not a multiple of the tile, odd image
    (1, 64, 8, 1, 1, 1, 1),          # one pixel
    (2, 128, 64, 3, 2, 7, 9),        # stride 2 on odd sizes
    # channel counts that are not multiples of 64: every 16-byte slot decodes its own filter tap, K padded with zero weights
    (2, 80, 80, 3, 2, 112, 160),
    (2, 160, 320, 3, 1, 56, 80),
    (2, 160, 64, 1, 1, 112, 160),
    (1, 480, 960, 3, 1, 14, 20),
    (2, 12, 80, 3, 1, 64, 96),       # the stem after ReOrg: 12 input channels, padded to 16
    (2, 320, 255, 1, 1, 56, 80),     # a Detect head: 255 outputs, written as 256
    (1, 8, 8, 3, 1, 5, 5),
]


def reference(x, w, b, k, s, act):
    y = torch.nn.functional.conv2d(x.float(), w.float(), b.float() if b is not None else None, stride=s, padding=k // 2)
    return torch.nn.functional.silu(y) if act == "silu" else y


@pytest.mark.parametrize("B,cin,cout,k,s,H,W", CASES)
def test_conv_nhwc_matches_torch_f32(gpu_device, B, cin, cout, k, s, H, W):
    from vlfm_amd.vlm import det_ops

    g = torch.Generator(device="cpu").manual_seed(B * 1000 + cin + cout + k + s + H)
    x = (torch.randn(B, cin, H, W, generator=g) * 0.7).half().to(gpu_device).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).half().to(gpu_device)
    b = (torch.randn(cout, generator=g) * 0.3).half().to(gpu_device)
    assert det_ops.conv_nhwc_supported(cin, cout, k, s)
    rows, bias = det_ops.pack_conv_weight(w, b)
    cin_p = (cin + 7) // 8 * 8
    xin = x if cin_p == cin else torch.nn.functional.pad(x, (0, 0, 0, 0, 0, cin_p - cin))
    for act in ("silu", None):
        y = det_ops.conv_nhwc(xin, rows, bias, k, s, act)
        ref = reference(x, w, b, k, s, act)
        assert y.shape[1] == (cout + 7) // 8 * 8 and y.is_contiguous(memory_format=torch.channels_last)
        y = y[:, :cout]
        assert y.shape == ref.shape
        err = (y.float() - ref).abs()
        tol = 2e-3 * ref.abs().clamp(min=1.0)
        assert bool((err <= tol).all()), f"max err {float(err.max()):.4g} at ref {float(ref.flatten()[err.argmax()]):.4g}"
    y = det_ops.conv_nhwc(xin, rows, None, k, s, None)[:, :cout]     # no bias
    ref = reference(x, w, None, k, s, None)
    assert bool(((y.float() - ref).abs() <= 2e-3 * ref.abs().clamp(min=1.0)).all())


def test_conv_nhwc_reads_and_writes_channel_slices(gpu_device):
    """Pixel strides are arguments: the input may be a channel slice of a wider NHWC buffer and the output may be written into a
    slice of a concatenation buffer (what removes the torch.cat copies of the ELAN blocks); the rest of the buffer is untouched."""
    from vlfm_amd.vlm import det_ops

    g = torch.Generator(device="cpu").manual_seed(5)
    B, H, W = 2, 20, 24
    wide = (torch.randn(B, 192, H, W, generator=g)).half().to(gpu_device).contiguous(memory_format=torch.channels_last)
    x = wide[:, 64:128]                                   # channels 64..127 of every pixel
    w = (torch.randn(128, 64, 3, 3, generator=g) / 24).half().to(gpu_device)
    b = torch.randn(128, generator=g).half().to(gpu_device)
    cat = torch.full((B, 320, H, W), 7.0, dtype=torch.float16, device=gpu_device).contiguous(memory_format=torch.channels_last)
    out = det_ops.conv_nhwc(x, *det_ops.pack_conv_weight(w, b), 3, 1, "silu", out=cat[:, 64:192])
    assert out.data_ptr() == cat[:, 64:192].data_ptr()
    ref = reference(x, w, b, 3, 1, "silu")
    assert bool(((cat[:, 64:192].float() - ref).abs() <= 2e-3 * ref.abs().clamp(min=1.0)).all())
    assert bool((cat[:, :64] == 7.0).all()) and bool((cat[:, 192:] == 7.0).all())


def test_conv_nhwc_converts_nchw_input_and_refuses_unsupported_shapes(gpu_device):
    from vlfm_amd import _lib
    from vlfm_amd.vlm import det_ops

    g = torch.Generator(device="cpu").manual_seed(6)
    x = torch.randn(1, 64, 12, 12, generator=g).half().to(gpu_device)              # NCHW memory: converted (one pass)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24).half().to(gpu_device)
    y = det_ops.conv_nhwc(x, det_ops.pack_conv_weight(w)[0], None, 3, 1, None)
    ref = reference(x, w, None, 3, 1, None)
    assert bool(((y.float() - ref).abs() <= 2e-3 * ref.abs().clamp(min=1.0)).all())
    assert det_ops.conv_nhwc_supported(80, 80, 3, 1) and det_ops.conv_nhwc_supported(64, 255, 1, 1)
    assert not det_ops.conv_nhwc_supported(64, 64, 5, 1) and not det_ops.conv_nhwc_supported(64, 64, 3, 3)
    assert not det_ops.conv_nhwc_supported(64, 64, 3, 1, padding=0) and not det_ops.conv_nhwc_supported(64, 64, 3, 1, groups=2)
    z = torch.zeros(64, dtype=torch.float16, device=gpu_device)
    rc = _lib.lib().vlfm_conv_nhwc_f16(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), z.data_ptr(), 1, 12, 12, 60, 64, 3, 1, 64,
                                       64, 0, None)
    assert rc == _lib.VLFM_ERR_INVALID


def test_e6e_graph_on_hip_convolutions_matches_the_framework_path(gpu_device):
    """The whole yolov7-e6e graph (262 modules, here at a quarter of the width so that it runs in seconds) with EVERY convolution
    on csrc/conv_nhwc.hip against the same f16 weights on the framework's convolutions, both judged against the f32 evaluation
    of the same network: the four feature maps entering Detect and the decoded predictions.  Both f16 paths round after every one
    of ~100 layers with different summation orders, so they differ from each other by a few per cent of the largest feature;
    the bound written here: the HIP path is no further from f32 than 1.5 x the framework's f16 path (+ 0.5 % of the largest
    magnitude)."""
    import copy

    from vlfm_amd.vlm.yolov7_e6e import Conv, Detect, YoloV7E6E

    torch.manual_seed(11)
    with torch.device(gpu_device):
        net = YoloV7E6E(width_multiple=0.25)
    net.init_random(11).eval()
    x = torch.rand(2, 3, 128, 192, device=gpu_device).half()

    # a random-init net this deep shrinks its activations into f16 denormals: rescale every convolution once (data-dependent, on
    # this input) so that its pre-activation has unit variance and the comparison below is about O(1) numbers
    def unit_variance(conv, inp, out):
        s = out.float().std().clamp(min=1e-12)
        conv.weight.div_(s)
        return out / s

    hooks = [m.conv.register_forward_hook(unit_variance) for m in net.modules() if isinstance(m, Conv)]
    with torch.no_grad():
        net(x.float())
    for h in hooks:
        h.remove()
    net.fuse_()
    for m in net.modules():      # f16-representable weights in all three evaluations
        if isinstance(m, torch.nn.Conv2d):
            m.weight.data = m.weight.data.half().float()
    feats = {}

    def run(model, inp):
        det = next(m for m in model.modules() if isinstance(m, Detect))
        hook = det.register_forward_hook(lambda m, i, o: feats.__setitem__("cur", [t.float().clone() for t in i[0]]))
        with torch.inference_mode():
            pred = model(inp).float()
        hook.remove()
        return feats["cur"], pred

    f32_feats, f32_pred = run(copy.deepcopy(net).float(), x.float())
    net.half()
    lib_feats, lib_pred = run(net, x)
    n = net.use_hip_conv_()
    assert n == sum(isinstance(m, Conv) for m in net.modules()) + 4 == 244      # every convolution taken
    hip_feats, hip_pred = run(net, x.contiguous(memory_format=torch.channels_last))
    assert len(f32_feats) == len(hip_feats) == 4
    for ref, lib, hip in zip(f32_feats + [f32_pred], lib_feats + [lib_pred], hip_feats + [hip_pred]):
        assert ref.shape == hip.shape and float(ref.abs().max()) > 0.5      # O(1) features, not denormals
        e_lib, e_hip = float((lib - ref).abs().max()), float((hip - ref).abs().max())
        assert e_hip <= 1.5 * e_lib + 0.005 * float(ref.abs().max()), (e_hip, e_lib, float(ref.abs().max()))
        # and in the mean the two f16 paths are equally close to f32
        m_lib, m_hip = float((lib - ref).abs().mean()), float((hip - ref).abs().mean())
        assert m_hip <= 1.5 * m_lib + 1e-4, (m_hip, m_lib)


def test_predict_batch_slices_large_batches(gpu_device):
    """YOLOv7.predict_batch sends batches beyond MAX_FRAMES_PER_FORWARD through the network in slices (the convolution kernel
    addresses with 32-bit element offsets): same detections as one forward."""
    from vlfm_amd.vlm.yolov7 import YOLOv7

    det = YOLOv7(device=gpu_device, allow_random_init=True, width_multiple=0.25)
    assert det.hip_convs == 244
    with torch.no_grad():   # make the random net emit confident boxes
        for m in det.model.model[-1].m:
            m.bias.fill_(0.0)
            m.bias.view(3, 85)[:, 4] = 1.5
        det.model.model[-1].use_hip_conv_()          # re-pack the head's filter rows with the new bias
    g = torch.Generator().manual_seed(9)
    frames = torch.randint(0, 255, (5, 480, 640, 3), generator=g, dtype=torch.uint8).to(gpu_device)
    whole = det.predict_batch(frames)
    det.MAX_FRAMES_PER_FORWARD = 2
    sliced = det.predict_batch(frames)
    assert len(whole) == len(sliced) == 5 and sum(d.num_detections for d in whole) > 0
    for a, b in zip(whole, sliced):
        assert a.num_detections == b.num_detections and torch.equal(a.boxes, b.boxes) and torch.equal(a.logits, b.logits)
        assert a.phrases == b.phrases


@pytest.mark.parametrize("B,C,H,W", [(2, 80, 224, 320), (3, 8, 2, 2), (1, 320, 14, 20)])
def test_maxpool2x2_nhwc_is_the_framework_pooling(gpu_device, B, C, H, W):
    from vlfm_amd.vlm import det_ops

    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g).half().to(gpu_device).contiguous(memory_format=torch.channels_last)
    got = det_ops.maxpool2x2_nhwc(x)
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, torch.nn.functional.max_pool2d(x, 2, 2))                 # a maximum has no rounding
    odd = torch.randn(1, 12, 5, 6).half().to(gpu_device)                             # not a shape the kernel takes: framework path
    assert torch.equal(det_ops.maxpool2x2_nhwc(odd), torch.nn.functional.max_pool2d(odd, 2, 2))

"""Host side of csrc/conv_nhwc.hip (no GPU): the filter-row packing the kernel reads (det_ops.pack_conv_weight), the layer shapes
it takes, the in-place concatenation plan of the yolov7-e6e graph, and the tile-shape estimate (vlfm_conv_nhwc_tile).  The
convolution itself is checked on the GPU in tests/test_conv_nhwc_gpu.py; the reference side is vlfm/vlm/yolov7.py:35-48,89."""
import ctypes

import pytest
import torch

from vlfm_amd import _lib
from vlfm_amd.vlm import det_ops
from vlfm_amd.vlm.yolov7_e6e import Concat, Conv, YoloV7E6E


def conv_from_rows(x, rows, bias, cin, cout, k, stride):
    """The implicit GEMM the kernel computes, spelled with unfold on the CPU: out[pixel][n] = sum_q patch[pixel][q] * rows[n][q]
    with q running over (ky, kx, channel) of a zero-padded NHWC patch."""
    B, cin_p, H, W = x.shape
    pad = k // 2
    cols = torch.nn.functional.unfold(x.float(), k, padding=pad, stride=stride)              # [B, cin_p*k*k, L], (c, ky, kx) order
    L = cols.shape[-1]
    cols = cols.view(B, cin_p, k * k, L).permute(0, 3, 2, 1).reshape(B, L, k * k * cin_p)   # -> (ky, kx, c) order
    ktot = k * k * cin_p
    out = cols @ rows[:, :ktot].float().t() + (bias.float() if bias is not None else 0.0)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    return out.view(B, Ho, Wo, -1).permute(0, 3, 1, 2)[:, :cout]


@pytest.mark.parametrize("cin,cout,k,stride", [(64, 64, 3, 1), (12, 80, 3, 1), (80, 80, 3, 2), (320, 255, 1, 1), (8, 8, 3, 1),
                                                (160, 64, 1, 1)])
def test_packed_filter_rows_are_the_implicit_gemm_operand(cin, cout, k, stride):
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    w = torch.randn(cout, cin, k, k, generator=g).half()
    b = torch.randn(cout, generator=g).half()
    rows, bias = det_ops.pack_conv_weight(w, b)
    cin_p, cout_p = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
    ktot = k * k * cin_p
    assert rows.dtype == torch.float16 and rows.is_contiguous() and rows.shape == (cout_p, (ktot + 63) // 64 * 64)
    assert bias.shape == (cout_p,) and bool((bias[cout:] == 0).all()) and bool((rows[cout:] == 0).all())
    assert bool((rows[:, ktot:] == 0).all())                                  # K padding multiplies whatever the kernel fetches by 0
    x = torch.randn(2, cin, 9, 11, generator=g).half()
    xp = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, cin_p - cin))
    ref = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), stride=stride, padding=k // 2)
    got = conv_from_rows(xp, rows, bias, cin, cout, k, stride)
    assert torch.allclose(got, ref, atol=1e-3, rtol=1e-3)
    assert det_ops.pack_conv_weight(w)[1] is None


def test_supported_layer_shapes():
    ok = det_ops.conv_nhwc_supported
    assert ok(64, 64, 3, 1) and ok(80, 80, (3, 3), (2, 2), (1, 1)) and ok(12, 80, 3, 1) and ok(320, 255, 1, 1, 0)
    assert not ok(64, 64, 5, 1) and not ok(64, 64, 3, 3) and not ok(64, 64, 3, 1, padding=0) and not ok(64, 64, (3, 1), 1)
    assert not ok(64, 64, 3, 1, groups=64) and not ok(64, 64, 3, 1, dilation=(2, 2))


def test_every_convolution_of_the_e6e_graph_is_a_supported_shape_and_concats_are_planned_in_place():
    net = YoloV7E6E().eval()
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
    assert len(convs) == 244
    assert all(det_ops.conv_nhwc_supported(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, c.groups, c.dilation)
               for c in convs)
    for m in net.modules():                    # what use_hip_conv_() sets on the GPU; the plan itself is host logic
        if isinstance(m, Conv):
            m.hip_conv = True
    net._plan_concats()
    concats = [i for i, m in enumerate(net.model) if isinstance(m, Concat)]
    assert set(net.cat_plan) <= set(concats) and len(net.cat_plan) == 25
    placed = [(i, j, off, c) for i, plan in net.cat_plan.items() for j, off, c, here in plan if here]
    assert len(placed) == 149 and sum(len(p) for p in net.cat_plan.values()) == 152
    assert len({j for _, j, _, _ in placed}) == len(placed)                   # a producer writes into ONE buffer
    for i, plan in net.cat_plan.items():
        offs = [off for _, off, _, _ in plan]
        assert offs == sorted(offs) and offs[0] == 0                          # slices tile the buffer in torch.cat's order
        assert all(a + c == b for (_, a, c, _), b in zip(plan, offs[1:] + [sum(c for _, _, c, _ in plan)]))
        for j, off, c, here in plan:
            assert j < i
            if here:
                assert net.out_slot[j] == (i, off, sum(c for _, _, c, _ in plan)) and off % 8 == 0 and c % 8 == 0


def tile(pixels, cin, cout, k):
    bm, bn = ctypes.c_int(), ctypes.c_int()
    _lib.check(_lib.lib().vlfm_conv_nhwc_tile(pixels, cin, cout, k, ctypes.byref(bm), ctypes.byref(bn)), "conv_nhwc_tile")
    return bm.value, bn.value


def test_tile_shape_estimate():
    shapes = {(256, 256), (256, 128), (256, 64), (128, 128), (128, 64), (64, 128), (64, 64)}
    for pixels in (1, 70, 4480, 35840, 1146880, 9175040):
        for cin, cout, k in ((64, 64, 3), (128, 128, 3), (512, 512, 3), (1280, 640, 1), (320, 255, 1), (80, 80, 1), (16, 80, 3)):
            assert tile(pixels, cin, cout, k) in shapes
    # many pixels, 64 channels: no tile wider than the layer (a 128-wide tile would spend half its MFMAs on padding)
    assert tile(128 * 112 * 160, 64, 64, 3)[1] == 64
    # a 7x10 map of a small batch: few pixels -> small pixel tiles, so that the 256 CUs all get work
    assert tile(8 * 70, 512, 512, 3)[0] <= 128
    # plenty of pixels and channels: never the smallest shape
    bm, bn = tile(128 * 28 * 40, 320, 640, 3)
    assert bm * bn >= 128 * 128
    assert _lib.lib().vlfm_conv_nhwc_tile(0, 64, 64, 3, None, None) == _lib.VLFM_ERR_INVALID

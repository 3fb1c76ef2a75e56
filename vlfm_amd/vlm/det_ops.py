"""Python faces of the detector-side HIP kernels (csrc/detect_ops.hip) + the yolov7 post-processing built on them."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib

IMAGENET_MEAN = (0.485, 0.456, 0.406)  # grounding_dino.py:54
IMAGENET_STD = (0.229, 0.224, 0.225)


def _stream() -> int:
    # (the raw handle of torch's current stream on the current device: torch.cuda.current_stream() builds a Stream object and looks the
    # device up twice -- 9 us per launch, 4.5 ms per step of a 64-environment full step)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


_AREA_TABS: Dict[Tuple[str, int, int], Tuple[torch.Tensor, torch.Tensor, torch.Tensor, int]] = {}


def _area_tab(device, ssize: int, dsize: int):
    key = (str(device), ssize, dsize)
    if key not in _AREA_TABS:
        cap = int(np.ceil(ssize / dsize)) + 2
        first, count = np.zeros(dsize, np.int32), np.zeros(dsize, np.int32)
        w = np.zeros((dsize, cap), np.float32)
        k = _lib.check(_lib.lib().vlfm_resize_area_tab_host(ssize, dsize, first.ctypes.data, count.ctypes.data,
                                                            w.ctypes.data, cap), "resize_area_tab_host")
        _AREA_TABS[key] = (torch.from_numpy(first).to(device), torch.from_numpy(count).to(device),
                           torch.from_numpy(w).to(device), cap)
        assert k <= cap
    return _AREA_TABS[key]


def resize_area(images_u8: torch.Tensor, out_h: int, out_w: int, dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """[n,H,W,3] u8 (device) -> cv2.resize(.., (out_w,out_h), INTER_AREA) -> CHW -> /255: [n,3,out_h,out_w]."""
    assert images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3
    images_u8 = images_u8.contiguous()
    n, H, W, _ = images_u8.shape
    dev = images_u8.device
    xf, xc, xw, xk = _area_tab(dev, W, out_w)
    yf, yc, yw, yk = _area_tab(dev, H, out_h)
    out = torch.empty((n, 3, out_h, out_w), dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().vlfm_resize_area_batched(images_u8.data_ptr(), n, H, W, out_h, out_w, xf.data_ptr(),
                                                      xc.data_ptr(), xw.data_ptr(), xk, yf.data_ptr(), yc.data_ptr(),
                                                      yw.data_ptr(), yk, out.data_ptr(),
                                                      {torch.float32: 0, torch.float16: 1}[dtype], _stream()),
                   "resize_area")
    return out


def to_tensor_normalize(images_u8: torch.Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """[n,H,W,3] u8 (device) -> torchvision to_tensor + normalize -> [n,3,H,W] f32."""
    assert images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3
    images_u8 = images_u8.contiguous()
    n, H, W, _ = images_u8.shape
    out = torch.empty((n, 3, H, W), dtype=torch.float32, device=images_u8.device)
    m, s = (ctypes.c_float * 3)(*mean), (ctypes.c_float * 3)(*std)
    with torch.cuda.device(images_u8.device):
        _lib.check(_lib.lib().vlfm_to_tensor_normalize_batched(images_u8.data_ptr(), n, H, W, ctypes.addressof(m),
                                                              ctypes.addressof(s), out.data_ptr(), _stream()),
                   "to_tensor_normalize")
    return out


def nms(boxes_xyxy: torch.Tensor, scores: torch.Tensor, iou_threshold: float, max_keep: Optional[int] = None) -> torch.Tensor:
    """torchvision.ops.nms: indices of the kept boxes, by descending score.  Device-resident result (int64)."""
    assert boxes_xyxy.is_cuda and boxes_xyxy.dtype == torch.float32 and boxes_xyxy.shape[-1] == 4
    n = boxes_xyxy.shape[0]
    dev = boxes_xyxy.device
    if n == 0:
        return torch.zeros(0, dtype=torch.int64, device=dev)
    boxes_xyxy = boxes_xyxy.contiguous()
    order = torch.sort(scores, descending=True).indices.to(torch.int32).contiguous()
    max_keep = n if max_keep is None else min(max_keep, n)
    scratch = torch.empty(_lib.lib().vlfm_nms_scratch_bytes(n), dtype=torch.uint8, device=dev)
    keep = torch.empty(max_keep, dtype=torch.int32, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().vlfm_nms(boxes_xyxy.data_ptr(), order.data_ptr(), n, float(iou_threshold),
                                      scratch.data_ptr(), scratch.numel(), keep.data_ptr(), num.data_ptr(), max_keep,
                                      _stream()), "nms")
    return keep[: int(num.item())].to(torch.int64)


def xywh2xyxy(x: torch.Tensor) -> torch.Tensor:
    y = x.clone()
    y[..., 0] = x[..., 0] - x[..., 2] / 2
    y[..., 1] = x[..., 1] - x[..., 3] / 2
    y[..., 2] = x[..., 0] + x[..., 2] / 2
    y[..., 3] = x[..., 1] + x[..., 3] / 2
    return y


# what the last non_max_suppression calls worked on (host ints that the function reads back anyway; bench.py reports them)
NMS_STATS = {"frames": 0, "candidates": 0, "boxes_in": 0}


def non_max_suppression(prediction: torch.Tensor, conf_thres: float = 0.25, iou_thres: float = 0.45,
                        classes: Optional[Sequence[int]] = None, agnostic: bool = False,
                        nms_fn=nms) -> List[torch.Tensor]:
    """yolov7 utils.general.non_max_suppression [ext] as the reference calls it (yolov7.py:91-97: multi_label off,
    no labels): per image an (k,6) tensor (x1, y1, x2, y2, conf, cls).  ``nms_fn`` is the HIP NMS on the GPU.

    Same arithmetic per image as the published loop; what is batched is the candidate selection: ONE objectness mask -> index
    list -> per-image counts for the whole batch (two host round trips) instead of a masked gather per image (a device
    synchronisation each: 8 ms of the 128-frame step), and images without candidates cost nothing."""
    B = prediction.shape[0]
    nc = prediction.shape[2] - 5
    max_wh, max_det, max_nms = 4096, 300, 30000
    dev = prediction.device
    out = [torch.zeros((0, 6), device=dev)] * B
    cand = (prediction[..., 4] > conf_thres).nonzero()            # [(image, row)], image-major, rows ascending
    NMS_STATS["frames"] += B
    NMS_STATS["candidates"] += int(cand.shape[0])
    NMS_STATS["boxes_in"] = int(prediction.shape[1])
    if cand.shape[0] == 0:
        return out
    x = prediction[cand[:, 0], cand[:, 1]].clone()
    if nc == 1:
        x[:, 5:] = x[:, 4:5]
    else:
        x[:, 5:] *= x[:, 4:5]  # conf = obj_conf * cls_conf
    box = xywh2xyxy(x[:, :4])
    conf, j = x[:, 5:].max(1, keepdim=True)
    keep = conf.view(-1) > conf_thres
    if classes is not None:
        keep &= (j == torch.tensor(classes, device=dev)).any(1)
    x = torch.cat((box, conf, j.float()), 1)[keep]
    img = cand[:, 0][keep]
    counts = torch.bincount(img, minlength=B).cpu().tolist()      # one transfer for the whole batch
    start = 0
    for xi, n in enumerate(counts):
        if not n:
            continue
        xs = x[start:start + n]
        start += n
        if n > max_nms:
            xs = xs[xs[:, 4].argsort(descending=True)[:max_nms]]
        c = xs[:, 5:6] * (0 if agnostic else max_wh)  # class offset trick
        i = nms_fn((xs[:, :4] + c).float(), xs[:, 4].float(), iou_thres, max_det)
        out[xi] = xs[i]
    return out


def scale_coords(img1_shape, coords: torch.Tensor, img0_shape) -> torch.Tensor:
    """yolov7 utils.general.scale_coords [ext] + clip_coords."""
    gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
    pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    coords[:, 0].clamp_(0, img0_shape[1])
    coords[:, 1].clamp_(0, img0_shape[0])
    coords[:, 2].clamp_(0, img0_shape[1])
    coords[:, 3].clamp_(0, img0_shape[0])
    return coords


def ms_deform_attn(value: torch.Tensor, spatial_shapes_list, level_start_index: torch.Tensor,
                   sampling_locations: torch.Tensor, attention_weights: torch.Tensor) -> torch.Tensor:
    """HIP multi-scale deformable attention: value [B,S,heads,D], sampling_locations [B,Q,heads,L,P,2],
    attention_weights [B,Q,heads,L,P] (all f32, device) -> [B,Q,heads*D]."""
    B, S, heads, D = value.shape
    _, Q, _, Lv, Pn, _ = sampling_locations.shape
    dev = value.device
    shapes = torch.tensor([[int(h), int(w)] for h, w in spatial_shapes_list], dtype=torch.int32, device=dev)
    out = torch.empty((B, Q, heads * D), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().vlfm_ms_deform_attn(value.contiguous().data_ptr(), shapes.data_ptr(),
                                                  level_start_index.to(torch.int32).contiguous().data_ptr(),
                                                  sampling_locations.contiguous().data_ptr(),
                                                  attention_weights.contiguous().data_ptr(), B, Q, heads, D, Lv, Pn, S,
                                                  out.data_ptr(), _stream()), "ms_deform_attn")
    return out


def ms_deform_attn_fused(value: torch.Tensor, spatial_shapes_list, level_start_index: torch.Tensor, offsets_logits: torch.Tensor,
                         reference_points: torch.Tensor, n_levels: int, n_points: int) -> torch.Tensor:
    """The sampling with its softmax and location arithmetic inside the kernel (csrc/detect_ops.hip): value [B,S,8,32],
    offsets_logits [B,Q,8*L*P*3] (the raw output of the sampling_offsets | attention_weights Linears), reference_points [B,Q,L,2|4]
    -> [B,Q,256]."""
    B, S, heads, D = value.shape
    Q = offsets_logits.shape[1]
    C = reference_points.shape[-1]
    dev = value.device
    key = (str(dev), tuple((int(h), int(w)) for h, w in spatial_shapes_list))
    shapes = _SHAPE_CACHE.get(key)
    if shapes is None:
        shapes = _SHAPE_CACHE[key] = torch.tensor([[int(h), int(w)] for h, w in spatial_shapes_list], dtype=torch.int32, device=dev)
    ls = level_start_index if level_start_index.dtype == torch.int32 else level_start_index.to(torch.int32)
    out = torch.empty((B, Q, heads * D), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().vlfm_ms_deform_attn_fused(value.contiguous().data_ptr(), shapes.data_ptr(), ls.contiguous().data_ptr(),
                                                        offsets_logits.contiguous().data_ptr(), reference_points.contiguous().data_ptr(),
                                                        B, Q, heads, D, n_levels, n_points, C, S, out.data_ptr(), _stream()),
                   "ms_deform_attn_fused")
    return out


_SHAPE_CACHE: Dict[Tuple, torch.Tensor] = {}


def patch_hf_deformable_attention(model) -> int:
    """Route every HF ``MultiScaleDeformableAttention`` module of ``model`` (GroundingDINO) through the HIP kernel when its
    inputs are f32 device tensors; anything else keeps the module's own PyTorch path.  Returns the number patched."""
    n = 0
    for mod in model.modules():
        if type(mod).__name__ == "MultiScaleDeformableAttention":
            torch_forward = mod.forward

            def forward(value, value_spatial_shapes, value_spatial_shapes_list, level_start_index, sampling_locations,
                        attention_weights, im2col_step, _fallback=torch_forward):
                if value.is_cuda and value.dtype == torch.float32 and sampling_locations.dtype == torch.float32:
                    return ms_deform_attn(value, value_spatial_shapes_list, level_start_index, sampling_locations,
                                          attention_weights.to(torch.float32))
                return _fallback(value, value_spatial_shapes, value_spatial_shapes_list, level_start_index,
                                 sampling_locations, attention_weights, im2col_step)

            mod.forward = forward
            n += 1
    return n


_ZERO_PAGES: Dict[str, torch.Tensor] = {}


def conv_nhwc_supported(cin: int, cout: int, ksize, stride, padding=None, groups: int = 1, dilation=(1, 1)) -> bool:
    """The layer shapes csrc/conv_nhwc.hip takes (vlfm_conv_nhwc_f16) once ``pack_conv_weight`` has padded the channel counts to
    multiples of 8: square 1x1 / 3x3 filters with 'same' padding, stride 1 or 2, no groups, no dilation."""
    k = ksize if isinstance(ksize, int) else (ksize[0] if ksize[0] == ksize[1] else -1)
    s = stride if isinstance(stride, int) else (stride[0] if stride[0] == stride[1] else -1)
    pad_ok = padding is None or padding in (k // 2, (k // 2, k // 2))
    return k in (1, 3) and s in (1, 2) and pad_ok and groups == 1 and tuple(dilation) == (1, 1) and cin > 0 and cout > 0


def pack_conv_weight(weight: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """[Cout, Cin, k, k] (+ bias) -> the operands of ``conv_nhwc``: f16 filter rows [Cout_p, Kpad] holding each filter's
    [k][k][Cin_p] weights (the channels_last order) followed by zeros up to Kpad = round_up(k * k * Cin_p, 64), and the f16 bias
    [Cout_p]; Cin_p / Cout_p = the channel counts rounded up to multiples of 8 with zero weights (the 12-channel stem reads a
    16-channel input, the 255-channel Detect heads write 256 channels of which the caller keeps 255)."""
    cout, cin, k, k2 = weight.shape
    assert k == k2
    cin_p, cout_p = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
    w = torch.zeros((cout_p, k, k, cin_p), dtype=torch.float16, device=weight.device)
    w[:cout, :, :, :cin] = weight.detach().permute(0, 2, 3, 1).to(torch.float16)
    ktot = k * k * cin_p
    kpad = (ktot + 63) // 64 * 64
    rows = torch.zeros((cout_p, kpad), dtype=torch.float16, device=weight.device)
    rows[:, :ktot] = w.reshape(cout_p, ktot)
    b = None
    if bias is not None:
        b = torch.zeros(cout_p, dtype=torch.float16, device=weight.device)
        b[:cout] = bias.detach().to(torch.float16)
    return rows, b


def conv_nhwc(x: torch.Tensor, w_rows: torch.Tensor, bias: Optional[torch.Tensor], ksize: int, stride: int = 1,
              act: Optional[str] = "silu", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(conv2d(x, w, padding=ksize // 2, stride) + bias) on the matrix cores (csrc/conv_nhwc.hip): one convolution of the
    yolov7-e6e graph (vlfm/vlm/yolov7.py:89) as an implicit GEMM, bias + SiLU in the epilogue.
      x       [B, Cin, H, W] f16 whose MEMORY is NHWC: a channels_last tensor or a channel slice of one (anything else is
              converted, which costs a pass); Cin % 8 == 0 (``pack_conv_weight`` pads the filter; pad the input alike)
      w_rows, bias   from ``pack_conv_weight``
      out     optional [B, Cout_p, Ho, Wo] f16 view with NHWC memory (e.g. a channel slice of a concatenation buffer)
    Returns a channels_last [B, Cout_p, Ho, Wo] tensor (or ``out``)."""
    assert x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and w_rows.dtype == torch.float16 and w_rows.is_contiguous()
    B, cin, H, W = x.shape
    cout, k = w_rows.shape[0], int(ksize)
    assert cin % 8 == 0 and cout % 8 == 0 and w_rows.shape[1] == (k * k * cin + 63) // 64 * 64, (tuple(w_rows.shape), cin, k)
    assert bias is None or (bias.dtype == torch.float16 and bias.is_contiguous() and bias.numel() == cout)

    def nhwc_pixel_stride(t: torch.Tensor) -> int:
        b, c, h, w = t.shape
        ps = t.stride(3) if w > 1 else (t.stride(2) if h > 1 else (t.stride(0) if b > 1 else c))
        ok = ((c == 1 or t.stride(1) == 1) and (w == 1 or t.stride(3) == ps) and (h == 1 or t.stride(2) == w * ps)
              and (b == 1 or t.stride(0) == h * w * ps) and ps >= c and ps % 8 == 0 and t.data_ptr() % 16 == 0)
        return ps if ok else -1

    xp = nhwc_pixel_stride(x)
    if xp < 0:
        x = x.contiguous(memory_format=torch.channels_last)
        xp = nhwc_pixel_stride(x)
        assert xp > 0
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if out is None:
        out = torch.empty((B, cout, Ho, Wo), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    op = nhwc_pixel_stride(out)
    assert out.shape == (B, cout, Ho, Wo) and out.dtype == torch.float16 and op > 0, "out must be an NHWC-memory f16 view"
    key = str(x.device)
    if key not in _ZERO_PAGES:
        _ZERO_PAGES[key] = torch.zeros(64, dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().vlfm_conv_nhwc_f16(x.data_ptr(), w_rows.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                 out.data_ptr(), _ZERO_PAGES[key].data_ptr(), B, H, W, cin, cout, k, int(stride),
                                                 xp, op, {None: 0, "silu": 1}[act], _stream()), "conv_nhwc_f16")
    return out


def maxpool2x2_nhwc(x: torch.Tensor) -> torch.Tensor:
    """``F.max_pool2d(x, 2, 2)`` of a channels_last f16 tensor with even sides and C % 8 == 0 (csrc/conv_nhwc.hip): DownC's
    pooling branch at HBM rate; anything else goes to the framework."""
    B, C, H, W = x.shape
    if not (x.is_cuda and x.dtype == torch.float16 and C % 8 == 0 and H % 2 == 0 and W % 2 == 0
            and x.is_contiguous(memory_format=torch.channels_last)):
        return torch.nn.functional.max_pool2d(x, 2, 2)
    out = torch.empty((B, C, H // 2, W // 2), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().vlfm_maxpool2x2_nhwc_f16(x.data_ptr(), out.data_ptr(), B, H, W, C, _stream()), "maxpool2x2_nhwc_f16")
    return out


def fold_batchnorm_(model) -> int:
    """Inference-time folding of every eval-mode BatchNorm2d that directly follows a Conv2d inside an ``nn.Sequential`` (the
    conv + bn + SiLU triples of the YOLOv7-class network, TinyViT's ``Conv2d_BN``): the convolution gets the scaled weights,
    the BatchNorm becomes an ``ops.BiasAct`` holding the folded bias -- and the SiLU / GELU module behind it, if there is one --
    i.e. what yolov7's ``fuse()`` / TinyViT's ``Conv2d_BN.fuse()`` do before deployment, with bias + activation as ONE in-place
    pass over the activation instead of the framework's BatchNorm pass + activation pass (MIOpen convolutions take no bias).
    Same function up to float rounding.  Call AFTER the checkpoint is loaded.  Returns the number folded."""
    import torch.nn as nn
    from torch.nn.utils.fusion import fuse_conv_bn_eval

    from .ops import BiasAct

    n = 0
    for m in list(model.modules()):
        if not isinstance(m, nn.Sequential):
            continue
        names = list(m._modules.keys())
        for pos, (a, b) in enumerate(zip(names, names[1:])):
            conv, bn = m._modules[a], m._modules[b]
            if isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and not bn.training:
                fused = fuse_conv_bn_eval(conv, bn)
                bias = fused.bias.detach().clone()
                fused.bias = None
                act = None
                if pos + 2 < len(names):
                    nxt = m._modules[names[pos + 2]]
                    if isinstance(nxt, nn.SiLU):
                        act = "silu"
                    elif isinstance(nxt, nn.GELU) and getattr(nxt, "approximate", "none") == "none":
                        act = "gelu"
                    if act is not None:
                        m._modules[names[pos + 2]] = nn.Identity()
                m._modules[a] = fused
                m._modules[b] = BiasAct(bias, act, inplace=True)   # the fused conv's output has this one consumer
                n += 1
    return n


def patch_convs_as_gemm(model) -> int:
    """Evaluate the GEMM-shaped convolutions of ``model`` (GroundingDINO: the Swin patch embedding, kernel == stride, and the
    1x1 ``input_proj`` convolutions) as one matrix multiplication each.  For these f32 shapes MIOpen falls back to its
    ``naive_conv`` kernel (13 % of the detector's GPU time in rocprofv3); a patch / pointwise convolution IS a GEMM over
    the unfolded pixels, and hipBLASLt runs it at the f32 MFMA rate.  Same arithmetic up to the summation order of the
    reduction (checked against nn.Conv2d in tests/test_detect_cpu.py).  Returns the number of modules patched."""
    import torch.nn as nn
    import torch.nn.functional as F

    n = 0
    for mod in model.modules():
        if not isinstance(mod, nn.Conv2d) or mod.groups != 1 or mod.dilation != (1, 1) or mod.padding not in ((0, 0), "valid"):
            continue
        kh, kw = mod.kernel_size
        if mod.stride != (kh, kw) and not (kh == kw == 1 and mod.stride == (1, 1)):
            continue

        def forward(x, _m=mod, _kh=kh, _kw=kw):
            b, c, h, w = x.shape
            oh, ow = h // _kh, w // _kw
            if _kh == 1 and _kw == 1:
                rows = x.permute(0, 2, 3, 1).reshape(b * h * w, c)
            else:   # non-overlapping patches: [B, C, oh, kh, ow, kw] -> rows of (c, kh, kw), the weight's own order
                rows = x[:, :, :oh * _kh, :ow * _kw].reshape(b, c, oh, _kh, ow, _kw).permute(0, 2, 4, 1, 3, 5) \
                    .reshape(b * oh * ow, c * _kh * _kw)
            y = F.linear(rows, _m.weight.reshape(_m.out_channels, -1), _m.bias)
            return y.view(b, oh, ow, _m.out_channels).permute(0, 3, 1, 2)

        mod.forward = forward
        n += 1
    return n


def cache_text_branch(model) -> None:
    """GroundingDINO's caption is constant for an episode (the class list), yet HF's forward re-runs BERT on it for every
    frame.  Memoise ``model.model.text_backbone`` per caption batch: the caller announces the captions it is about to run
    through ``model.vlfm_text_key`` (a hashable, host-side value -- no device read-back, so the lookup is legal inside a
    HIP-graph capture); without a key the backbone runs as usual."""
    backbone = model.model.text_backbone
    plain = backbone.forward
    cache = {}
    model.vlfm_text_key = None

    def forward(input_ids, attention_mask=None, token_type_ids=None, position_ids=None, **kw):
        key = model.vlfm_text_key
        if key is None or torch.is_grad_enabled():
            return plain(input_ids, attention_mask, token_type_ids, position_ids, **kw)
        key = (key, str(input_ids.device), tuple(input_ids.shape), tuple(sorted(kw.items())))
        if key not in cache:
            if len(cache) > 64:
                cache.clear()
            cache[key] = plain(input_ids, attention_mask, token_type_ids, position_ids, **kw)
        return cache[key]

    backbone.forward = forward


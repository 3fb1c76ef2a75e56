"""Python wrappers of the VLM-side HIP kernels (csrc/vlm_ops.hip)."""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from .. import _lib

# LAVIS blip2 eval image processor = CLIP statistics
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_coeff_cache: Dict[Tuple[str, int, int], Tuple[torch.Tensor, torch.Tensor, int]] = {}


def _stream() -> int:
    # (the raw handle of torch's current stream on the current device: torch.cuda.current_stream() builds a Stream object and looks the
    # device up twice -- 9 us per launch, 4.5 ms per step of a 64-environment full step)
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def resample_coeffs(in_size: int, out_size: int, bilinear: bool = False) -> Tuple[np.ndarray, np.ndarray, int]:
    """Pillow bicubic (or bilinear) taps for one axis (host): bounds [out,2] int32, kk [out,ksize] int32, ksize."""
    bounds = np.zeros((out_size, 2), np.int32)
    cap = out_size * (2 * int(np.ceil(2.0 * max(1.0, in_size / out_size))) + 1)
    kk = np.zeros(cap, np.int32)
    ks = ctypes.c_int(0)
    _lib.check(_lib.lib().vlfm_resample_coeffs_filter_host(in_size, out_size, int(bilinear), bounds.ctypes.data,
                                                           kk.ctypes.data, cap, ctypes.byref(ks)),
               "resample_coeffs_host")
    return bounds, kk[: out_size * ks.value].reshape(out_size, ks.value), ks.value


def _device_coeffs(device, in_size: int, out_size: int, bilinear: bool = False):
    key = (str(device), in_size, out_size, bilinear)
    if key not in _coeff_cache:
        b, k, ks = resample_coeffs(in_size, out_size, bilinear)
        _coeff_cache[key] = (torch.from_numpy(b).to(device), torch.from_numpy(np.ascontiguousarray(k)).to(device), ks)
    return _coeff_cache[key]


def preprocess_rgb(images_u8: torch.Tensor, out_size: int = 224, dtype: torch.dtype = torch.float16,
                   mean=CLIP_MEAN, std=CLIP_STD, patch_size: int = 0) -> torch.Tensor:
    """[n,H,W,3] uint8 (device) -> [n,3,out,out] normalised, via the PIL-exact bicubic kernels.  With ``patch_size`` P
    the same values come out as [n,(out/P)^2,3*P*P]: the im2col rows of the ViT's stride-P patch-embedding conv."""
    assert images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3
    images_u8 = images_u8.contiguous()
    n, H, W, _ = images_u8.shape
    dev = images_u8.device
    hb, hk, hks = _device_coeffs(dev, W, out_size)
    vb, vk, vks = _device_coeffs(dev, H, out_size)
    tmp = torch.empty((n, H, out_size, 3), dtype=torch.uint8, device=dev)
    if patch_size:
        g = out_size // patch_size
        out = torch.empty((n, g * g, 3 * patch_size * patch_size), dtype=dtype, device=dev)
    else:
        out = torch.empty((n, 3, out_size, out_size), dtype=dtype, device=dev)
    m = (ctypes.c_float * 3)(*mean)
    s = (ctypes.c_float * 3)(*std)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().vlfm_preprocess_rgb_batched(images_u8.data_ptr(), n, H, W, out_size, hb.data_ptr(),
                                                         hk.data_ptr(), hks, vb.data_ptr(), vk.data_ptr(), vks,
                                                         ctypes.addressof(m), ctypes.addressof(s), tmp.data_ptr(),
                                                         out.data_ptr(), _DTYPE_CODE[dtype], int(patch_size), _stream()),
                   "preprocess_rgb")
    return out


def itc_head(query_feats: torch.Tensor, proj_t: torch.Tensor, proj_bias: torch.Tensor,
             text_feats: torch.Tensor) -> torch.Tensor:
    """query_feats [B,NQ,H] f32, proj_t [H,P] f32, proj_bias [P], text_feats [B,P] -> [B] cosines.
    The 768->256 projection is one hipBLASLt GEMM over all B*NQ rows; normalise + dot + max is the HIP epilogue."""
    B, NQ, Hd = query_feats.shape
    P = proj_t.shape[1]
    for t in (query_feats, proj_t, proj_bias, text_feats):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    assert proj_t.shape[0] == Hd and text_feats.shape == (B, P)
    proj = torch.addmm(proj_bias, query_feats.view(B * NQ, Hd), proj_t)
    out = torch.empty(B, dtype=torch.float32, device=query_feats.device)
    with torch.cuda.device(query_feats.device):
        _lib.check(_lib.lib().vlfm_itc_head_batched(proj.data_ptr(), B, NQ, P, text_feats.data_ptr(), out.data_ptr(),
                                                   _stream()), "itc_head")
    return out


SAM_MEAN = (123.675, 116.28, 103.53)   # Sam.pixel_mean / pixel_std [ext segment_anything / mobile_sam build_sam]
SAM_STD = (58.395, 57.12, 57.375)


def sam_target_size(h: int, w: int, long_side: int = 1024) -> Tuple[int, int]:
    """ResizeLongestSide.get_preprocess_shape [ext]."""
    scale = long_side * 1.0 / max(h, w)
    return int(h * scale + 0.5), int(w * scale + 0.5)


def preprocess_sam(images_u8: torch.Tensor, long_side: int = 1024) -> Tuple[torch.Tensor, Tuple[int, int]]:
    """[n,H,W,3] u8 (device) -> SamPredictor.set_image's network input [n,3,1024,1024] f32 (PIL BILINEAR resize of the
    longest side to 1024, (x-mean)/std, zero pad) and the resized (h, w)."""
    assert images_u8.is_cuda and images_u8.dtype == torch.uint8 and images_u8.dim() == 4 and images_u8.shape[-1] == 3
    images_u8 = images_u8.contiguous()
    n, H, W, _ = images_u8.shape
    oh, ow = sam_target_size(H, W, long_side)
    dev = images_u8.device
    hb, hk, hks = _device_coeffs(dev, W, ow, True)
    vb, vk, vks = _device_coeffs(dev, H, oh, True)
    tmp = torch.empty((n, H, ow, 3), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 3, long_side, long_side), dtype=torch.float32, device=dev)
    m, s = (ctypes.c_float * 3)(*SAM_MEAN), (ctypes.c_float * 3)(*SAM_STD)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().vlfm_preprocess_sam_batched(images_u8.data_ptr(), n, H, W, oh, ow, hb.data_ptr(),
                                                         hk.data_ptr(), hks, vb.data_ptr(), vk.data_ptr(), vks,
                                                         ctypes.addressof(m), ctypes.addressof(s), long_side,
                                                         tmp.data_ptr(), out.data_ptr(), _stream()), "preprocess_sam")
    return out, (oh, ow)


def layernorm_bias(x: torch.Tensor, channel_bias, weight: torch.Tensor, bias: torch.Tensor, eps: float) -> torch.Tensor:
    """LayerNorm(x + channel_bias) over the last dimension of an f16 tensor (HIP kernel, f32 statistics);
    ``channel_bias`` is an f32 [dim] constant or None."""
    assert x.is_cuda and x.dtype == torch.float16 and x.is_contiguous() and weight.dtype == torch.float16
    dim = x.shape[-1]
    y = torch.empty_like(x)
    _lib.check(_lib.lib().vlfm_layernorm_bias_f16(x.data_ptr(), channel_bias.data_ptr() if channel_bias is not None else None,
                                                  weight.data_ptr(), bias.data_ptr(), y.data_ptr(), x.numel() // dim, dim,
                                                  float(eps), _stream()), "layernorm_bias_f16")
    return y


_ACT_CODE = {None: 0, "none": 0, "gelu": 1, "silu": 2}


def bias_act_(y: torch.Tensor, bias: torch.Tensor, act=None) -> torch.Tensor:
    """In place on a contiguous NCHW f32 / f16 tensor: y = act(y + bias[c]); act in (None, "gelu", "silu").  One pass instead
    of the framework's two (csrc/detect_ops.hip)."""
    assert y.is_cuda and y.dim() == 4 and y.is_contiguous() and y.dtype in (torch.float32, torch.float16)
    assert bias.dtype == y.dtype and bias.numel() == y.shape[1] and bias.is_contiguous()
    n, c, h, w = y.shape
    _lib.check(_lib.lib().vlfm_bias_act_nchw(y.data_ptr(), bias.data_ptr(), n, c, h * w, 0 if y.dtype == torch.float32 else 1,
                                             _ACT_CODE[act], _stream()), "bias_act_nchw")
    return y


_BIAS_ACT_MAX_PLANES = 65535 * 16   # vlfm_bias_act_nchw's grid: n * C planes per launch


class BiasAct(torch.nn.Module):
    """``act(x + bias[c])`` as a module: takes the place of a folded BatchNorm2d (+ the activation module behind it).
    ``inplace=True`` overwrites its input (one pass of the HIP kernel): right exactly where ``det_ops.fold_batchnorm_``
    installs it -- behind a convolution whose output has no other consumer.  The default leaves the input alone."""

    def __init__(self, bias: torch.Tensor, act=None, inplace: bool = False):
        super().__init__()
        self.bias = torch.nn.Parameter(bias.detach().clone(), requires_grad=False)
        self.act = act
        self.inplace = inplace

    def forward(self, x):
        return bias_act(x, self.bias, self.act, inplace=self.inplace)


def bias_act(x: torch.Tensor, bias: torch.Tensor, act=None, inplace: bool = False) -> torch.Tensor:
    """act(x + bias[c]).  ``inplace=True``: the HIP kernel, OVERWRITING ``x`` (a contiguous NCHW f32 / f16 tensor on the GPU
    without grad, within the kernel's plane limit); otherwise -- and for anything the kernel does not take -- the framework
    ops, which leave ``x`` untouched."""
    if (inplace and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.float16) and x.is_contiguous()
            and not x.requires_grad and x.shape[0] * x.shape[1] <= _BIAS_ACT_MAX_PLANES):
        return bias_act_(x, bias if bias.dtype == x.dtype else bias.to(x.dtype), act)
    y = x + bias.to(x.dtype).view(1, -1, 1, 1)
    if act == "gelu":
        return torch.nn.functional.gelu(y)
    if act == "silu":
        return torch.nn.functional.silu(y)
    return y


def depthwise_conv3x3(x: torch.Tensor, weight: torch.Tensor, bias, stride: int = 1, gelu: bool = False) -> torch.Tensor:
    """Depthwise 3x3 convolution (padding 1, stride 1 or 2) of an NCHW f32 tensor, + bias, + exact GELU: the HIP kernel behind
    TinyViT's depthwise convolutions (csrc/detect_ops.hip).  weight [C,1,3,3]."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
    n, c, h, w = x.shape
    assert weight.shape == (c, 1, 3, 3) and weight.dtype == torch.float32 and weight.is_contiguous()
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().vlfm_dwconv3x3_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                             y.data_ptr(), n, c, h, w, int(stride), int(gelu), _stream()), "dwconv3x3_f32")
    return y


def layernorm_rows(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, window: int = 0, shift: int = 0,
                   pad_zero: bool = False) -> torch.Tensor:
    """LayerNorm over the channels of a contiguous f32 [B, H, W, C] tensor (csrc/sam_ops.hip).  ``window == 0``: same shape.
    ``window > 0``: TinyViTBlock's pad + window partition + ``attn.norm`` in one pass -> [B * nWy * nWx, window^2, C], the rows
    of the zero-padded image in window order (padded positions hold LayerNorm(0) = bias, as in the reference).  ``shift`` /
    ``pad_zero``: Swin's form -- norm, zero padding AFTER the norm, roll by -shift, partition (SwinLayer.forward [ext])."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
    B, H, W, C = x.shape
    assert weight.dtype == torch.float32 and bias.dtype == torch.float32 and weight.numel() == C == bias.numel()
    if window > 0:
        nwy, nwx = (H + window - 1) // window, (W + window - 1) // window
        out = torch.empty((B * nwy * nwx, window * window, C), dtype=torch.float32, device=x.device)
    else:
        out = torch.empty_like(x)
    _lib.check(_lib.lib().vlfm_layernorm_rows_shifted_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, C,
                                                          int(window), float(eps), int(shift), int(pad_zero), _stream()),
               "layernorm_rows_f32")
    return out


def window_reverse_add_(x: torch.Tensor, windows: torch.Tensor, window: int, shift: int = 0) -> torch.Tensor:
    """In place: x[b, y, x] += windows[row of (b, y, x) in window order] -- TinyViTBlock's window reverse + crop + residual add
    (csrc/sam_ops.hip).  x [B, H, W, C] f32 contiguous, windows [B * nWy * nWx, window^2, C]."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
    B, H, W, C = x.shape
    nwy, nwx = (H + window - 1) // window, (W + window - 1) // window
    assert windows.is_contiguous() and windows.dtype == torch.float32 and windows.numel() == B * nwy * nwx * window * window * C
    _lib.check(_lib.lib().vlfm_window_reverse_add_shifted_f32(x.data_ptr(), windows.data_ptr(), B, H, W, C, int(window), int(shift),
                                                              _stream()), "window_reverse_add_f32")
    return x


def depthwise_conv3x3_nhwc(x: torch.Tensor, w9c: torch.Tensor, bias) -> torch.Tensor:
    """Depthwise 3x3 convolution (stride 1, padding 1) + bias of a contiguous f32 [B, H, W, C] tensor; ``w9c`` is the [C, 1, 3, 3]
    weight as [9, C] (``weight.view(C, 9).t().contiguous()``)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
    B, H, W, C = x.shape
    assert w9c.shape == (9, C) and w9c.is_contiguous() and w9c.dtype == torch.float32
    out = torch.empty_like(x)
    _lib.check(_lib.lib().vlfm_dwconv3x3_nhwc_f32(x.data_ptr(), w9c.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                  out.data_ptr(), B, H, W, C, _stream()), "dwconv3x3_nhwc_f32")
    return out


def window_attention(qkv: torch.Tensor, bias_t: torch.Tensor, heads: int, scale: float, mask_t=None) -> torch.Tensor:
    """TinyViT's window attention (head width 32) in one kernel (csrc/sam_ops.hip): ``qkv`` [windows, tokens, heads * 96] f32 as the
    qkv Linear emits it (per head q | k | v), ``bias_t`` [heads, tokens, tokens] the additive bias transposed over its last two
    dimensions; returns softmax(scale * q k^T + bias) v as [windows, tokens, heads * 32]."""
    assert qkv.is_cuda and qkv.dtype == torch.float32 and qkv.dim() == 3 and qkv.is_contiguous() and qkv.shape[2] == heads * 96
    nw, n, _ = qkv.shape
    assert bias_t.shape == (heads, n, n) and bias_t.is_contiguous() and bias_t.dtype == torch.float32 and n <= 256
    out = torch.empty((nw, n, heads * 32), dtype=torch.float32, device=qkv.device)
    per_image = 1
    if mask_t is not None:   # Swin's shifted-window mask [windows per image, tokens, tokens] (symmetric)
        assert mask_t.dim() == 3 and mask_t.shape[1:] == (n, n) and mask_t.is_contiguous() and mask_t.dtype == torch.float32
        per_image = mask_t.shape[0]
        assert nw % per_image == 0
    _lib.check(_lib.lib().vlfm_window_attention_masked_f32(qkv.data_ptr(), bias_t.data_ptr(),
                                                           mask_t.data_ptr() if mask_t is not None else None, per_image,
                                                           out.data_ptr(), nw, n, heads, float(scale), _stream()),
               "window_attention_f32")
    return out


GEMM_F16_EPILOGUE = {"bias": 0, "bias_gelu": 1, "accumulate": 2}


def linear_f16_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Shapes csrc/gemm_f16.hip takes: f16 on the GPU, K % 64 == 0, N % 8 == 0, each operand below 4 GB."""
    k = weight.shape[1]
    return (x.is_cuda and x.dtype == torch.float16 and weight.dtype == torch.float16 and x.shape[-1] == k and k % 64 == 0
            and weight.shape[0] % 8 == 0 and x.numel() * 2 < (1 << 32) and weight.numel() * 2 < (1 << 32))


def linear_f16(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: str = "bias",
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """epilogue(x @ weight.T + bias), f16 in / f16 out, f32 accumulation on the matrix cores: the hand-written 256 x 256 x 64
    8-phase GEMM of csrc/gemm_f16.hip.  x [M, K], weight [N, K], bias [N] or None.  epilogue "bias": the plain Linear;
    "bias_gelu": exact (erf) GELU on the f32 accumulators (hipBLASLt only fuses the tanh approximation, so the library path is a GEMM
    plus a separate pass over the 4x-wide activation); "accumulate": ``out += x @ weight.T (+ bias)`` in place, summed in f32 -- the
    residual-stream GEMMs (``out`` is required)."""
    assert x.is_cuda and x.dtype == torch.float16 and x.is_contiguous() and x.dim() == 2
    assert weight.dtype == torch.float16 and weight.is_contiguous() and weight.shape[1] == x.shape[1]
    assert bias is None or (bias.dtype == torch.float16 and bias.is_contiguous())
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        assert epilogue != "accumulate", "accumulate needs the tensor to accumulate into"
        out = torch.empty((M, N), dtype=torch.float16, device=x.device)
    else:
        assert out.dtype == torch.float16 and out.is_contiguous() and out.shape == (M, N) and out.device == x.device
    _lib.check(_lib.lib().vlfm_gemm_f16_nt(x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                           out.data_ptr(), M, N, K, GEMM_F16_EPILOGUE[epilogue], _stream()), "gemm_f16_nt")
    return out


def linear_gelu(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """gelu(x @ weight.T + bias), exact (erf) form: ``linear_f16(..., epilogue="bias_gelu")``."""
    return linear_f16(x, weight, bias, "bias_gelu")


VIT_ATTENTION_TOKENS, VIT_ATTENTION_HEADS = 257, (88,)


def vit_attention(qkv: torch.Tensor, batch: int, tokens: int, heads: int, head_dim: int, scale: float) -> torch.Tensor:
    """softmax(q k^T * scale) v for the ViT-g shape (257 tokens, heads of 88): ``qkv`` is the qkv GEMM's output
    [batch*tokens, 3*heads*head_dim] f16 as it is; returns [batch*tokens, heads*head_dim] f16 for the projection GEMM."""
    assert qkv.is_cuda and qkv.dtype == torch.float16 and qkv.is_contiguous()
    assert qkv.numel() == batch * tokens * 3 * heads * head_dim
    out = torch.empty((batch * tokens, heads * head_dim), dtype=torch.float16, device=qkv.device)
    _lib.check(_lib.lib().vlfm_vit_attention_f16(qkv.data_ptr(), out.data_ptr(), batch, tokens, heads, head_dim,
                                                 float(scale), _stream()), "vit_attention_f16")
    return out


# ------------------------------------------------------------------------------------------------ f32 Linear on the matrix cores
_ACT32 = {None: 0, "none": 0, "relu": 1, "gelu": 2}
GEMM_F32_PRECISION = {"exact": 0, "split": 1}
_overflow_flags: Dict[tuple, torch.Tensor] = {}
_split_cache: Dict[int, tuple] = {}


def gemm_f32_overflow_flag(device, owner: str = "") -> torch.Tensor:
    """The sticky device int the split-precision GEMMs OR into when an operand leaves f16's range: one per device and ``owner`` (each
    network reads and clears its own, so one network's fallback never hides another's overflow)."""
    d = torch.device(device)
    # "cuda" (the reference's spelling) and "cuda:0" are the same device: the kernels are handed the flag of ``x.device`` (always
    # indexed), the networks look theirs up with whatever the caller passed -- both must meet in one tensor
    index = d.index if d.index is not None else (torch.cuda.current_device() if d.type == "cuda" else 0)
    key = (d.type, index, owner)
    if key not in _overflow_flags:
        _overflow_flags[key] = torch.zeros(1, dtype=torch.int32, device=torch.device(d.type, index))
    return _overflow_flags[key]


def tensor_version(t: torch.Tensor) -> int:
    """``t._version``, or 0 for a tensor created under ``torch.inference_mode()`` (those do not track in-place updates -- and cannot
    be updated in place outside inference mode either)."""
    try:
        return t._version
    except RuntimeError:
        return 0


def split_weight(weight: torch.Tensor, owner: str = "") -> Tuple[torch.Tensor, torch.Tensor]:
    """(hi, lo') f16 planes of an f32 weight: hi = f16(w), lo' = f16((w - hi) 2^11).  Memoised per tensor OBJECT (a weak reference
    proves it is still the same tensor: an address alone is reused by the allocator) and per ``_version`` (in-place updates).
    The range check (|w| >= 65504, inf, NaN) runs only when the planes are built, so its verdict is KEPT with the planes: a private
    device int per weight, OR-ed into the owner's flag on every use of the cached planes (a cleared owner flag does not forget it)."""
    import weakref

    key = id(weight)
    hit = _split_cache.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == tensor_version(weight) and hit[2] == weight.data_ptr():
        if hit[6]:     # known-bad planes (established at a synchronisation point): say so again, every time
            gemm_f32_overflow_flag(weight.device, owner).bitwise_or_(hit[5])
        return hit[3], hit[4]
    if len(_split_cache) > 2048:
        for k in [k for k, v in _split_cache.items() if v[0]() is None]:
            del _split_cache[k]
    w = weight.detach().contiguous()
    hi = torch.empty(w.shape, dtype=torch.float16, device=w.device)
    lo = torch.empty_like(hi)
    bad = torch.zeros(1, dtype=torch.int32, device=w.device)
    _lib.check(_lib.lib().vlfm_split_f32_to_f16_pair(w.data_ptr(), hi.data_ptr(), lo.data_ptr(), w.numel(), bad.data_ptr(),
                                                     _stream()), "split_f32")
    gemm_f32_overflow_flag(w.device, owner).bitwise_or_(bad)
    _split_cache[key] = [weakref.ref(weight), tensor_version(weight), weight.data_ptr(), hi, lo, bad, False, owner]
    return hi, lo


def split_weights_bad(device, owner: str = "") -> bool:
    """True when a cached hi/lo pair of ``owner`` was built from a weight outside f16's range.  One read-back over all of the owner's
    private flags: called after an overflow was seen (a synchronisation point), never per step.  Marks such entries so that every later
    use of their planes raises the owner's flag again."""
    d = torch.device(device)
    entries = [v for v in _split_cache.values() if v[7] == owner and v[0]() is not None and v[5].device.type == d.type
               and (d.index is None or v[5].device.index == d.index)]
    if not entries:
        return False
    verdict = torch.cat([v[5] for v in entries]).cpu().tolist()
    for v, b in zip(entries, verdict):
        v[6] = bool(b)
    return any(verdict)


def linear_f32_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Shapes csrc/gemm_f32.hip takes: f32 on the GPU, K % 32 == 0, enough rows and columns to fill a 128 x 128 tile halfway."""
    k = weight.shape[1]
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and k % 32 == 0 and x.shape[-1] == k
            and weight.shape[0] >= 32 and x.numel() // k >= 64 and not torch.is_grad_enabled())


def linear_f32(x: torch.Tensor, weight: torch.Tensor, bias=None, act=None, residual=None, precision: str = "exact",
               out=None, owner: str = "") -> torch.Tensor:
    """act(x @ weight.T + bias) + residual for f32 tensors on the matrix cores (csrc/gemm_f32.hip); x [..., K], weight [N, K].
    precision "exact": v_mfma_f32_32x32x2_f32 (a k-ordered f32 fma chain); "split": the f16 hi/lo form (f32-grade, 5.3x the rate;
    check ``gemm_f32_overflow_flag`` at the next synchronisation point)."""
    K = weight.shape[1]
    N = weight.shape[0]
    lead = x.shape[:-1]
    x2 = x.reshape(-1, K)
    if not x2.is_contiguous() or x2.data_ptr() % 16:
        x2 = x2.contiguous()
    w = weight if weight.is_contiguous() else weight.contiguous()
    M = x2.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    res = None
    if residual is not None:
        res = residual.reshape(M, N)
        if not res.is_contiguous() or res.data_ptr() % 16:
            res = res.contiguous()
    b = None
    if bias is not None:
        b = bias if bias.is_contiguous() else bias.contiguous()
    prec = GEMM_F32_PRECISION[precision]
    hi = lo = None
    if prec == 1:
        hi, lo = split_weight(w, owner)
    _lib.check(_lib.lib().vlfm_gemm_f32_nt(x2.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None,
                                           res.data_ptr() if res is not None else None, out.data_ptr(), M, N, K, _ACT32[act], prec,
                                           hi.data_ptr() if hi is not None else None, lo.data_ptr() if lo is not None else None,
                                           gemm_f32_overflow_flag(x.device, owner).data_ptr(), _stream()), "gemm_f32_nt")
    return out.view(*lead, N)


_tuned_gemms_state = None


def use_tuned_gemms() -> bool:
    """Library GEMMs (the ViT's qkv / projection / fc2 go to hipBLASLt) through the solutions PyTorch's TunableOp measured to be the
    fastest per shape on this image: ``vlfm_amd/tunableop_results.csv``, written by ``tools/tune_gemms.py`` (no tuning happens at
    run time; shapes that are not in the file, or a file recorded by other library versions, take the library's own heuristic).
    ``VLFM_TUNED_GEMMS=0`` switches it off."""
    global _tuned_gemms_state
    if _tuned_gemms_state is None:
        import os

        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tunableop_results.csv")
        ok = False
        if os.environ.get("VLFM_TUNED_GEMMS", "1") != "0" and os.path.exists(path) and torch.cuda.is_available():
            try:
                import torch.cuda.tunable as tunable

                import tempfile

                tunable.enable(True)
                tunable.tuning_enable(False)
                # TunableOp writes its table back to ITS file name when the process exits: that must not be the package's file
                tunable.set_filename(os.path.join(tempfile.gettempdir(), f"vlfm_amd_tunableop_{os.getpid()}.csv"))
                ok = bool(tunable.read_file(path))
            except Exception as exc:  # noqa: BLE001 -- an optional speed-up: say so and go on with the library's heuristic
                import warnings

                warnings.warn(f"tuned GEMM solutions not loaded ({type(exc).__name__}: {exc})")
                ok = False
        _tuned_gemms_state = ok
    return _tuned_gemms_state

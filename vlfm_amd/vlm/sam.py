"""vlfm.vlm.sam, MI355X in-process (reference: /root/reference/vlfm/vlm/sam.py:24-69).

``MobileSAM.segment_bbox(image, bbox)`` = ``SamPredictor.set_image`` + ``predict(box=..., multimask_output=False)[0]`` of the
un-vendored mobile_sam package [ext]: longest side -> 1024 (PIL bilinear), (x-mean)/std, zero pad to 1024^2, TinyViT image
encoder -> 256x64x64 embedding, box prompt -> prompt encoder -> two-way mask decoder -> low-res mask, bilinear upsample to
1024^2, crop to the resized extent, bilinear to the original size, threshold at 0.  Returns the FULL-FRAME boolean mask
(the reference docstring says "cropped"; it is not -- SURVEY.md App. C8).

Network: TinyViT-5M encoder (written here from the published architecture: embed dims 64/128/160/320, depths 2/2/6/2,
heads 2/4/5/10, windows 7/7/14/7, MBConv stem stage, SAM neck) + HF ``transformers`` SAM prompt encoder / mask decoder
(same design as SAM ViT-*).  ``mobile_sam.pt`` loads through :func:`load_mobile_sam_state_dict` (strict key map, tested
on a synthetic state dict with the checkpoint's names and shapes -- the file itself is not available offline);
preprocessing is the Pillow-exact HIP resampler of csrc/vlm_ops.hip."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class _ConvBN(nn.Sequential):
    """conv (no bias) + BatchNorm; sub-module names ``c`` / ``bn`` as in the MobileSAM checkpoint."""

    def __init__(self, a: int, b: int, ks: int = 1, stride: int = 1, pad: int = 0, groups: int = 1):
        super().__init__()
        self.add_module("c", nn.Conv2d(a, b, ks, stride, pad, groups=groups, bias=False))
        self.add_module("bn", nn.BatchNorm2d(b))


def _folded(cb: "_ConvBN") -> bool:
    return isinstance(cb.bn, ops.BiasAct)


def _conv_act(cb: "_ConvBN", x: torch.Tensor, gelu: bool) -> torch.Tensor:
    """``gelu(cb(x))`` / ``cb(x)``; with the BatchNorm folded (det_ops.fold_batchnorm_) the bias and the GELU are one pass."""
    if _folded(cb):
        return ops.bias_act(cb.c(x), cb.bn.bias, "gelu" if gelu else None, inplace=True)   # the conv output is ours alone
    y = cb(x)
    return F.gelu(y) if gelu else y


def _dwconv(cb: "_ConvBN", x: torch.Tensor, gelu: bool) -> torch.Tensor:
    """A depthwise 3x3 ``_ConvBN`` (+ GELU): the HIP kernel once the BatchNorm is folded and the tensor lives on the GPU in f32
    (MIOpen falls back to its naive convolution for these: 13 ms of the 128-env full step), the framework path otherwise."""
    c = cb.c
    if (_folded(cb) and x.is_cuda and x.dtype == torch.float32 and c.groups == c.in_channels == c.out_channels
            and c.kernel_size == (3, 3) and c.padding == (1, 1) and c.stride[0] == c.stride[1] and c.stride[0] in (1, 2)
            and x.shape[3] % 4 == 0 and ((x.shape[3] - 1) // c.stride[0] + 1) % 4 == 0
            and (c.stride[0] == 1 or x.shape[3] % 2 == 0)):
        return ops.depthwise_conv3x3(x.contiguous(), c.weight, cb.bn.bias, c.stride[0], gelu)
    return _conv_act(cb, x, gelu)


class _PatchEmbed(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.seq = nn.Sequential(_ConvBN(3, dim // 2, 3, 2, 1), nn.GELU(), _ConvBN(dim // 2, dim, 3, 2, 1))

    def forward(self, x):
        return self.seq(x)


class _MBConv(nn.Module):
    def __init__(self, dim: int, expand: float = 4.0):
        super().__init__()
        h = int(dim * expand)
        self.conv1, self.conv2, self.conv3 = _ConvBN(dim, h), _ConvBN(h, h, 3, 1, 1, h), _ConvBN(h, dim)

    def forward(self, x):
        return F.gelu(x + self.conv3(_dwconv(self.conv2, _conv_act(self.conv1, x, True), True)))


class _PatchMerging(nn.Module):
    def __init__(self, dim: int, out: int):
        super().__init__()
        stride = 1 if out in (320, 448, 576) else 2  # MobileSAM keeps 64x64 from the 160 -> 320 merge on
        self.conv1, self.conv2, self.conv3 = _ConvBN(dim, out), _ConvBN(out, out, 3, stride, 1, out), _ConvBN(out, out)

    def forward(self, x):
        return self.conv3(_dwconv(self.conv2, _conv_act(self.conv1, x, True), True))


import threading as _threading

_PRECISION_LOCK = _threading.Lock()   # the switch below is module-wide: a fallback re-run holds this while it flips it
GEMM_PRECISION = "split"     # TinyViT's f32 Linears on the rows path: "split" (csrc/gemm_f32.hip, f16 operand pairs) | "library"


def _rows_linear(x: torch.Tensor, lin: nn.Linear, act=None, residual=None, out=None) -> torch.Tensor:
    """A Linear of TinyViT's rows path: the split-precision f32 GEMM with bias / exact GELU / residual in its epilogue when the shape
    allows (f32-grade results, ~2x the library's f32 rate; an operand beyond f16's range raises the "sam" overflow flag, which
    MobileSAM.segment_bbox turns into a repeat on the library GEMMs and MobileSAM.check_numerics into an error), else the framework."""
    if GEMM_PRECISION == "split" and ops.linear_f32_supported(x, lin.weight) and x.numel() // x.shape[-1] >= 1024:
        return ops.linear_f32(x, lin.weight, lin.bias, act=act, residual=residual, precision="split", out=out, owner="sam")
    y = lin(x)
    if act == "gelu":
        y = F.gelu(y)
    if residual is not None:
        y = y.add_(residual.reshape(y.shape))
    if out is not None:
        out.copy_(y.reshape(out.shape))
        return out.view(y.shape)
    return y


class _WindowAttention(nn.Module):
    def __init__(self, dim: int, heads: int, window: int):
        super().__init__()
        self.heads, self.kd = heads, dim // heads
        self.norm = nn.LayerNorm(dim)
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        pts = [(i, j) for i in range(window) for j in range(window)]
        offs: Dict[Any, int] = {}
        idx = []
        for p1 in pts:
            for p2 in pts:
                o = (abs(p1[0] - p2[0]), abs(p1[1] - p2[1]))
                idx.append(offs.setdefault(o, len(offs)))
        self.attention_biases = nn.Parameter(torch.zeros(heads, len(offs)))
        self.register_buffer("attention_bias_idxs", torch.tensor(idx).view(len(pts), len(pts)), persistent=False)

    def forward(self, x, normed: bool = False):  # [B*, N, C]; normed: ``x`` already went through self.norm
        b, n, c = x.shape
        if x.is_cuda and x.dtype == torch.float32 and self.kd == 32 and n <= 256 and not torch.is_grad_enabled():
            # one kernel per block: a lane per query, the window's key / value rows through scalar loads (csrc/sam_ops.hip)
            key = (str(x.device), ops.tensor_version(self.attention_biases), self.attention_biases.data_ptr())
            if getattr(self, "_bias_t", (None,))[0] != key:
                bias = self.attention_biases[:, self.attention_bias_idxs].detach().to(torch.float32)
                self._bias_t = (key, bias.transpose(1, 2).contiguous())
            a = ops.window_attention(_rows_linear(x if normed else self.norm(x), self.qkv), self._bias_t[1], self.heads, self.kd ** -0.5)
            return _rows_linear(a, self.proj)
        q, k, v = self.qkv(x if normed else self.norm(x)).view(b, n, self.heads, 3 * self.kd).split(self.kd, dim=3)
        bias = self.attention_biases[:, self.attention_bias_idxs].unsqueeze(0).to(x.dtype)
        a = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=bias)
        return self.proj(a.transpose(1, 2).reshape(b, n, c))


class _Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fc1, self.fc2 = nn.Linear(dim, hidden), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(self.norm(x))))


class _TinyViTBlock(nn.Module):
    def __init__(self, dim: int, heads: int, window: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.window = window
        self.attn = _WindowAttention(dim, heads, window)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.local_conv = _ConvBN(dim, dim, 3, 1, 1, dim)

    def forward(self, x):  # [B, C, H, W]
        b, c, h, w = x.shape
        ws = self.window
        t = x.permute(0, 2, 3, 1)
        ph, pw = (ws - h % ws) % ws, (ws - w % ws) % ws
        if ph or pw:
            t = F.pad(t, (0, 0, 0, pw, 0, ph))
        hp, wp = h + ph, w + pw
        t = t.view(b, hp // ws, ws, wp // ws, ws, c).transpose(2, 3).reshape(-1, ws * ws, c)
        t = self.attn(t).view(b, hp // ws, wp // ws, ws, ws, c).transpose(2, 3).reshape(b, hp, wp, c)[:, :h, :w]
        x = x + t.permute(0, 3, 1, 2)
        x = _dwconv(self.local_conv, x, False)
        t = x.permute(0, 2, 3, 1)
        t = t + self.mlp(t)
        return t.permute(0, 3, 1, 2)

    def rows_path(self, t: torch.Tensor) -> bool:
        """The NHWC-rows path (csrc/sam_ops.hip) takes f32 GPU tensors once the local_conv's BatchNorm is folded."""
        return (t.is_cuda and t.dtype == torch.float32 and _folded(self.local_conv) and t.shape[-1] % 4 == 0
                and not torch.is_grad_enabled() and not t.requires_grad)    # the rows kernels are forward-only and in place

    def forward_rows(self, t):  # [B, H, W, C] contiguous f32, OWNED by the caller's stage loop (updated in place)
        """The same block on NHWC rows: pad + window partition + attn.norm is one kernel, window reverse + crop + residual add is
        one (in place), the depthwise local_conv stays on NHWC rows, the MLP's norm is the row LayerNorm -- 12 passes over the
        activation instead of ~20, none of them strided."""
        b, h, w, c = t.shape
        ws = self.window
        a = self.attn(ops.layernorm_rows(t, self.attn.norm.weight, self.attn.norm.bias, self.attn.norm.eps, ws), normed=True)
        ops.window_reverse_add_(t, a.contiguous(), ws)
        lc = self.local_conv
        key = (str(t.device), ops.tensor_version(lc.c.weight), lc.c.weight.data_ptr())   # a weight reload must not meet a stale repack
        if getattr(self, "_w9c", (None,))[0] != key:
            self._w9c = (key, lc.c.weight.detach().reshape(c, 9).t().contiguous())
        t = ops.depthwise_conv3x3_nhwc(t, self._w9c[1], lc.bn.bias)
        m = self.mlp
        hid = _rows_linear(ops.layernorm_rows(t, m.norm.weight, m.norm.bias, m.norm.eps), m.fc1, act="gelu")
        t2 = t.view(-1, c)
        _rows_linear(hid.view(-1, hid.shape[-1]), m.fc2, residual=t2, out=t2)      # t += fc2(hid), in the GEMM's epilogue
        return t


class _Stage(nn.Module):
    """``blocks`` + optional ``downsample``: layers.0 is the MBConv stage, layers.1-3 the windowed-attention stages."""

    def __init__(self, blocks: List[nn.Module], downsample: Optional[nn.Module]):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)
        if downsample is not None:
            self.downsample = downsample

    def forward(self, x):
        blocks = list(self.blocks)
        if blocks and all(isinstance(b, _TinyViTBlock) for b in blocks):
            t = x.permute(0, 2, 3, 1)
            if all(b.rows_path(t) for b in blocks):      # one NCHW -> NHWC copy per stage instead of four per block
                t = t.contiguous()
                if t.data_ptr() == x.data_ptr():   # channels_last input: .contiguous() is a view, and the rows path writes in place
                    t = t.clone()
                for blk in blocks:
                    t = blk.forward_rows(t)
                x = t.permute(0, 3, 1, 2).contiguous()
                return self.downsample(x) if hasattr(self, "downsample") else x
        for blk in blocks:
            x = blk(x)
        return self.downsample(x) if hasattr(self, "downsample") else x


class _LayerNorm2d(nn.Module):
    def __init__(self, c: int, eps: float = 1e-6):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c)), eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        return self.weight[:, None, None] * ((x - u) / torch.sqrt(s + self.eps)) + self.bias[:, None, None]


class TinyViT(nn.Module):
    """TinyViT-5M as configured by MobileSAM (img 1024 -> 256 x 64 x 64).  Module names follow the ``image_encoder.*`` keys
    of ``mobile_sam.pt`` one to one (patch_embed.seq.N.{c,bn}, layers.N.blocks.M.{conv1..3 | attn,mlp,local_conv},
    layers.N.downsample.conv1..3, norm_head, head, neck.0-3), so the checkpoint's encoder loads by prefix stripping; the
    classification head (norm_head / head) is carried only because the checkpoint has it."""

    def __init__(self, dims=(64, 128, 160, 320), depths=(2, 2, 6, 2), heads=(2, 4, 5, 10), windows=(7, 7, 14, 7),
                 num_classes: int = 1000):
        super().__init__()
        self.patch_embed = _PatchEmbed(dims[0])
        layers: List[nn.Module] = [_Stage([_MBConv(dims[0]) for _ in range(depths[0])], _PatchMerging(dims[0], dims[1]))]
        for i in range(1, 4):
            layers.append(_Stage([_TinyViTBlock(dims[i], heads[i], windows[i]) for _ in range(depths[i])],
                                 _PatchMerging(dims[i], dims[i + 1]) if i < 3 else None))
        self.layers = nn.ModuleList(layers)
        self.norm_head = nn.LayerNorm(dims[3])
        self.head = nn.Linear(dims[3], num_classes)
        self.neck = nn.Sequential(nn.Conv2d(dims[3], 256, 1, bias=False), _LayerNorm2d(256),
                                  nn.Conv2d(256, 256, 3, padding=1, bias=False), _LayerNorm2d(256))

    def forward(self, x):
        x = self.patch_embed(x)
        for layer in self.layers:
            x = layer(x)
        return (self.neck(x),)


# mobile_sam.pt (segment-anything layout) -> transformers.SamModel names for the prompt encoder and the mask decoder
_SAM_RENAMES = (
    ("prompt_encoder.pe_layer.positional_encoding_gaussian_matrix", "prompt_encoder.shared_embedding.positional_embedding"),
    ("prompt_encoder.point_embeddings.", "prompt_encoder.point_embed."),
    ("prompt_encoder.mask_downscaling.0.", "prompt_encoder.mask_embed.conv1."),
    ("prompt_encoder.mask_downscaling.1.", "prompt_encoder.mask_embed.layer_norm1."),
    ("prompt_encoder.mask_downscaling.3.", "prompt_encoder.mask_embed.conv2."),
    ("prompt_encoder.mask_downscaling.4.", "prompt_encoder.mask_embed.layer_norm2."),
    ("prompt_encoder.mask_downscaling.6.", "prompt_encoder.mask_embed.conv3."),
    ("mask_decoder.output_upscaling.0.", "mask_decoder.upscale_conv1."),
    ("mask_decoder.output_upscaling.1.", "mask_decoder.upscale_layer_norm."),
    ("mask_decoder.output_upscaling.3.", "mask_decoder.upscale_conv2."),
    ("mask_decoder.transformer.norm_final_attn.", "mask_decoder.transformer.layer_norm_final_attn."),
    (".layers.0.", ".proj_in."), (".layers.1.", ".layers.0."), (".layers.2.", ".proj_out."),   # 3-layer MLP heads only
    (".norm1.", ".layer_norm1."), (".norm2.", ".layer_norm2."), (".norm3.", ".layer_norm3."), (".norm4.", ".layer_norm4."),
)


def mobile_sam_key_map(checkpoint_keys) -> Dict[str, List[str]]:
    """checkpoint key -> model keys it fills (``model`` = SamModel whose vision_encoder is :class:`TinyViT`)."""
    out: Dict[str, List[str]] = {}
    for k in checkpoint_keys:
        if k.startswith("image_encoder."):
            if k.endswith("attention_bias_idxs"):   # derived index table, rebuilt by the module
                continue
            out[k] = ["vision_encoder." + k[len("image_encoder."):]]
            continue
        t = k
        mlp_head = ("output_hypernetworks_mlps" in k) or ("iou_prediction_head" in k)
        for a, b in _SAM_RENAMES:
            if a.startswith(".layers.") and not mlp_head:
                continue
            if a in t:
                t = t.replace(a, b, 1)
                if a.startswith(".layers."):
                    break
        out[k] = [t]
        if t == "prompt_encoder.shared_embedding.positional_embedding":
            out[k].append("shared_image_embedding.positional_embedding")   # the same Gaussian matrix encodes the image grid
    return out


def build_mobile_sam_model():
    """transformers.SamModel (prompt encoder + two-way mask decoder: same design as SAM ViT-*) around TinyViT."""
    from transformers import SamConfig, SamModel

    cfg = SamConfig()
    cfg.vision_config.num_hidden_layers = 1  # the HF ViT encoder is replaced below; keep its construction cheap
    cfg.mask_decoder_config.layer_norm_eps = 1e-5  # segment-anything's decoder uses nn.LayerNorm's default, not HF's 1e-6
    model = SamModel(cfg)
    model.vision_encoder = TinyViT()
    return model.eval()


def load_mobile_sam_state_dict(model, state_dict) -> Dict[str, Any]:
    """Load ``mobile_sam.pt`` (a plain state dict of mobile_sam's ``Sam``: image_encoder.* TinyViT, prompt_encoder.*,
    mask_decoder.*) into ``model``.  Strict both ways: every checkpoint tensor must land on a model tensor of the same
    shape, and every model tensor must be filled -- anything else raises."""
    if "model" in state_dict and isinstance(state_dict["model"], dict):
        state_dict = state_dict["model"]
    own = model.state_dict()
    kmap = mobile_sam_key_map(state_dict.keys())
    filled, problems = set(), []
    with torch.no_grad():
        for k, targets in kmap.items():
            v = state_dict[k]
            for t in targets:
                if t not in own:
                    problems.append(f"{k} -> {t}: no such tensor in the model")
                elif tuple(own[t].shape) != tuple(v.shape):
                    problems.append(f"{k} -> {t}: shape {tuple(v.shape)} vs {tuple(own[t].shape)}")
                else:
                    own[t].copy_(v)
                    filled.add(t)
    missing = [t for t in own if t not in filled]
    if problems or missing:
        raise RuntimeError("mobile_sam checkpoint does not fit the model: " + "; ".join(problems[:8])
                           + (f"; model tensors left unfilled: {missing[:8]} (+{max(0, len(missing) - 8)})" if missing else ""))
    return {"tensors_loaded": len(kmap), "model_tensors_filled": len(filled)}


class MobileSAM:
    """sam.py:24-57.  ``sam_checkpoint`` is the reference's ``data/mobile_sam.pt``; a path that is given MUST load (a
    segmenter that silently falls back to random weights hands noise masks to the object map).  Benchmarks and shape
    tests opt into random weights explicitly with ``allow_random_init=True``."""

    def __init__(self, sam_checkpoint: Optional[str] = None, model_type: str = "vit_t", device: Optional[Any] = None,
                 seed: int = 0, allow_random_init: bool = False) -> None:
        import os

        from ..mapping.base_map import require_gpu

        self.device = require_gpu(device)
        assert model_type == "vit_t", "MobileSAM is the TinyViT ('vit_t') variant of SAM (sam.py:30)"
        sam_checkpoint = sam_checkpoint or os.environ.get("MOBILE_SAM_CHECKPOINT")   # sam.py:72-74 reads the same variable
        torch.manual_seed(seed)
        self.model = build_mobile_sam_model()
        if sam_checkpoint:
            if not os.path.isfile(sam_checkpoint):
                raise FileNotFoundError(f"MobileSAM checkpoint {sam_checkpoint!r} not found")
            report = load_mobile_sam_state_dict(self.model, torch.load(sam_checkpoint, map_location="cpu"))
            self.weights = f"{sam_checkpoint} ({report['tensors_loaded']} tensors)"
        elif allow_random_init:
            self.weights = "random-init (allow_random_init=True: benchmark / shape tests only)"
        else:
            raise ValueError("MobileSAM needs sam_checkpoint (the reference's data/mobile_sam.pt) or MOBILE_SAM_CHECKPOINT; "
                             "pass allow_random_init=True for a randomly initialised network (benchmarks only)")
        from .det_ops import fold_batchnorm_

        self.folded_batchnorms = fold_batchnorm_(self.model)   # TinyViT's Conv2d_BN.fuse(): conv + eval-mode BN = one conv
        self.model.to(self.device)
        self.mask_threshold = 0.0
        self.overflow_fallbacks = 0

    @torch.inference_mode()
    def _segment_bboxes_once(self, images_u8: torch.Tensor, boxes_xyxy: torch.Tensor) -> torch.Tensor:
        B, H, W, _ = images_u8.shape
        pix, (oh, ow) = ops.preprocess_sam(images_u8)
        emb = self.model.get_image_embeddings(pix)
        scale = torch.tensor([ow / W, oh / H, ow / W, oh / H], device=self.device, dtype=torch.float32)
        boxes = boxes_xyxy.to(self.device, torch.float32) * scale  # ResizeLongestSide.apply_boxes
        out = self.model(image_embeddings=emb, input_boxes=boxes, multimask_output=False)
        low = out.pred_masks[:, :, 0]                                   # [B,K,256,256]
        m = F.interpolate(low, (1024, 1024), mode="bilinear", align_corners=False)[..., :oh, :ow]
        m = F.interpolate(m, (H, W), mode="bilinear", align_corners=False)
        return m > self.mask_threshold

    def segment_bboxes(self, images_u8: torch.Tensor, boxes_xyxy: torch.Tensor) -> torch.Tensor:
        """images_u8 [B,H,W,3] u8 on device; boxes [B,K,4] pixel xyxy -> [B,K,H,W] bool masks.  With the split-precision GEMMs
        switched on, the overflow flag is read here (one 4-byte read-back; every caller reads the masks next anyway) and a flagged
        batch is repeated on the library's f32 GEMMs -- the batched callers (harness, clients) never see void masks."""
        global GEMM_PRECISION
        masks = self._segment_bboxes_once(images_u8, boxes_xyxy)
        if GEMM_PRECISION == "split":
            flag = ops.gemm_f32_overflow_flag(self.device, "sam")
            if int(flag.item()):
                flag.zero_()
                self.overflow_fallbacks += 1
                with _PRECISION_LOCK:
                    GEMM_PRECISION = "library"
                    try:
                        masks = self._segment_bboxes_once(images_u8, boxes_xyxy)
                    finally:
                        # a WEIGHT outside f16's range poisons its cached planes for good: stay on the library from then on
                        GEMM_PRECISION = "library" if ops.split_weights_bad(self.device, "sam") else "split"
        return masks

    def check_numerics(self) -> None:
        """Raise if a split-precision GEMM of the encoder met an operand outside f16's range since the last check (only reachable
        when something other than ``segment_bboxes`` ran the encoder)."""
        flag = ops.gemm_f32_overflow_flag(self.device, "sam")
        if int(flag.item()):
            flag.zero_()
            raise FloatingPointError("MobileSAM: an activation left f16's range inside a split-precision f32 GEMM; set "
                                     "vlfm_amd.vlm.sam.GEMM_PRECISION = 'library'")

    def segment_bbox(self, image: np.ndarray, bbox: List[int]) -> np.ndarray:
        img = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)[None]
        box = torch.tensor(bbox, dtype=torch.float32).view(1, 1, 4)
        return self.segment_bboxes(img, box)[0, 0].cpu().numpy()


class MobileSAMClient:
    """sam.py:60-69; ``port`` accepted and ignored (in-process).  ``emulate_jpeg=True`` reproduces the reference's
    quality-90 JPEG transport of the frame (server_wrapper.py:57-68) for A/B checks; the mask travels losslessly there too."""

    _shared: Dict[str, MobileSAM] = {}

    def __init__(self, port: int = 12183, device=None, emulate_jpeg: bool = False, **model_kwargs) -> None:
        key = str(device)
        if key not in MobileSAMClient._shared:
            MobileSAMClient._shared[key] = MobileSAM(device=device, **model_kwargs)
        self._model = MobileSAMClient._shared[key]
        self._emulate_jpeg = emulate_jpeg
        self.url = f"inprocess://mobile_sam (port {port} ignored)"

    def segment_bbox(self, image: np.ndarray, bbox: List[int]) -> np.ndarray:
        # the reference ships the mask as the bytes of a bool array reshaped to image.shape[:2] (sam.py:67)
        seen = image
        if self._emulate_jpeg:
            from .transport import jpeg_roundtrip

            seen = jpeg_roundtrip(image)
        return self._model.segment_bbox(seen, bbox).reshape(image.shape[:2])

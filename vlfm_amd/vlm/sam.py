"""vlfm.vlm.sam, MI355X in-process (reference: /root/reference/vlfm/vlm/sam.py:24-69).

``MobileSAM.segment_bbox(image, bbox)`` = ``SamPredictor.set_image`` + ``predict(box=..., multimask_output=False)[0]`` of the
un-vendored mobile_sam package [ext]: longest side -> 1024 (PIL bilinear), (x-mean)/std, zero pad to 1024^2, TinyViT image
encoder -> 256x64x64 embedding, box prompt -> prompt encoder -> two-way mask decoder -> low-res mask, bilinear upsample to
1024^2, crop to the resized extent, bilinear to the original size, threshold at 0.  Returns the FULL-FRAME boolean mask
(the reference docstring says "cropped"; it is not -- SURVEY.md App. C8).

Network: TinyViT-5M encoder (written here from the published architecture: embed dims 64/128/160/320, depths 2/2/6/2,
heads 2/4/5/10, windows 7/7/14/7, MBConv stem stage, SAM neck) + HF ``transformers`` SAM prompt encoder / mask decoder
(same design as SAM ViT-*).  ``mobile_sam.pt`` is not available offline -> random-init; preprocessing is the Pillow-exact HIP
resampler of csrc/vlm_ops.hip."""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class _ConvBN(nn.Sequential):
    def __init__(self, a: int, b: int, ks: int = 1, stride: int = 1, pad: int = 0, groups: int = 1):
        super().__init__(nn.Conv2d(a, b, ks, stride, pad, groups=groups, bias=False), nn.BatchNorm2d(b))


class _MBConv(nn.Module):
    def __init__(self, dim: int, expand: float = 4.0):
        super().__init__()
        h = int(dim * expand)
        self.conv1, self.conv2, self.conv3 = _ConvBN(dim, h), _ConvBN(h, h, 3, 1, 1, h), _ConvBN(h, dim)

    def forward(self, x):
        return F.gelu(x + self.conv3(F.gelu(self.conv2(F.gelu(self.conv1(x))))))


class _PatchMerging(nn.Module):
    def __init__(self, dim: int, out: int):
        super().__init__()
        stride = 1 if out in (320, 448, 576) else 2  # MobileSAM keeps 64x64 from the 160 -> 320 merge on
        self.conv1, self.conv2, self.conv3 = _ConvBN(dim, out), _ConvBN(out, out, 3, stride, 1, out), _ConvBN(out, out)

    def forward(self, x):
        return self.conv3(F.gelu(self.conv2(F.gelu(self.conv1(x)))))


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
        self.register_buffer("bias_idx", torch.tensor(idx).view(len(pts), len(pts)), persistent=False)

    def forward(self, x):  # [B*, N, C]
        b, n, c = x.shape
        q, k, v = self.qkv(self.norm(x)).view(b, n, self.heads, 3 * self.kd).split(self.kd, dim=3)
        bias = self.attention_biases[:, self.bias_idx].unsqueeze(0).to(x.dtype)
        a = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=bias)
        return self.proj(a.transpose(1, 2).reshape(b, n, c))


class _TinyViTBlock(nn.Module):
    def __init__(self, dim: int, heads: int, window: int, mlp_ratio: float = 4.0):
        super().__init__()
        self.window = window
        self.attn = _WindowAttention(dim, heads, window)
        self.local_conv = _ConvBN(dim, dim, 3, 1, 1, dim)
        self.mlp_norm = nn.LayerNorm(dim)
        self.fc1, self.fc2 = nn.Linear(dim, int(dim * mlp_ratio)), nn.Linear(int(dim * mlp_ratio), dim)

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
        x = self.local_conv(x)
        t = x.permute(0, 2, 3, 1)
        t = t + self.fc2(F.gelu(self.fc1(self.mlp_norm(t))))
        return t.permute(0, 3, 1, 2)


class _LayerNorm2d(nn.Module):
    def __init__(self, c: int, eps: float = 1e-6):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c)), eps

    def forward(self, x):
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        return self.weight[:, None, None] * ((x - u) / torch.sqrt(s + self.eps)) + self.bias[:, None, None]


class TinyViT(nn.Module):
    """TinyViT-5M as configured by MobileSAM (img 1024 -> 256 x 64 x 64)."""

    def __init__(self, dims=(64, 128, 160, 320), depths=(2, 2, 6, 2), heads=(2, 4, 5, 10), windows=(7, 7, 14, 7)):
        super().__init__()
        self.patch_embed = nn.Sequential(_ConvBN(3, dims[0] // 2, 3, 2, 1), nn.GELU(), _ConvBN(dims[0] // 2, dims[0], 3, 2, 1))
        stages: List[nn.Module] = [nn.Sequential(*[_MBConv(dims[0]) for _ in range(depths[0])]),
                                   _PatchMerging(dims[0], dims[1])]
        for i in range(1, 4):
            stages.append(nn.Sequential(*[_TinyViTBlock(dims[i], heads[i], windows[i]) for _ in range(depths[i])]))
            if i < 3:
                stages.append(_PatchMerging(dims[i], dims[i + 1]))
        self.stages = nn.Sequential(*stages)
        self.neck = nn.Sequential(nn.Conv2d(dims[3], 256, 1, bias=False), _LayerNorm2d(256),
                                  nn.Conv2d(256, 256, 3, padding=1, bias=False), _LayerNorm2d(256))

    def forward(self, x):
        return (self.neck(self.stages(self.patch_embed(x))),)


class MobileSAM:
    def __init__(self, sam_checkpoint: Optional[str] = None, model_type: str = "vit_t", device: Optional[Any] = None,
                 seed: int = 0) -> None:
        from transformers import SamConfig, SamModel

        from ..mapping.base_map import require_gpu

        self.device = require_gpu(device)
        torch.manual_seed(seed)
        cfg = SamConfig()
        cfg.vision_config.num_hidden_layers = 1  # the HF ViT encoder is replaced below; keep its construction cheap
        self.model = SamModel(cfg)
        self.model.vision_encoder = TinyViT()
        self.model.eval().to(self.device)
        self.weights = "random-init" if not sam_checkpoint else f"unavailable offline: {sam_checkpoint}"
        self.mask_threshold = 0.0

    @torch.inference_mode()
    def segment_bboxes(self, images_u8: torch.Tensor, boxes_xyxy: torch.Tensor) -> torch.Tensor:
        """images_u8 [B,H,W,3] u8 on device; boxes [B,K,4] pixel xyxy -> [B,K,H,W] bool masks."""
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

    def segment_bbox(self, image: np.ndarray, bbox: List[int]) -> np.ndarray:
        img = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)[None]
        box = torch.tensor(bbox, dtype=torch.float32).view(1, 1, 4)
        return self.segment_bboxes(img, box)[0, 0].cpu().numpy()


class MobileSAMClient:
    """sam.py:60-69; ``port`` accepted and ignored (in-process)."""

    _shared: Dict[str, MobileSAM] = {}

    def __init__(self, port: int = 12183, device=None, **model_kwargs) -> None:
        key = str(device)
        if key not in MobileSAMClient._shared:
            MobileSAMClient._shared[key] = MobileSAM(device=device, **model_kwargs)
        self._model = MobileSAMClient._shared[key]
        self.url = f"inprocess://mobile_sam (port {port} ignored)"

    def segment_bbox(self, image: np.ndarray, bbox: List[int]) -> np.ndarray:
        # the reference ships the mask as the bytes of a bool array reshaped to image.shape[:2] (sam.py:67)
        return self._model.segment_bbox(image, bbox).reshape(image.shape[:2])

"""vlfm.vlm.yolov7, MI355X in-process (reference: /root/reference/vlfm/vlm/yolov7.py:28-121).

``YOLOv7.predict(image)`` keeps the reference's pipeline -- cv2.resize to (640, 448) INTER_AREA, letterbox (a no-op at that
size), CHW, fp16, /255, model, non_max_suppression(0.25, 0.45), scale_coords + round, normalise by the ORIGINAL width and
height -- with the preprocessing and the NMS as HIP kernels (csrc/detect_ops.hip) and everything batched over the resident
environments (``predict_batch``).  ``YOLOv7Client`` keeps the client signature; the model lives in this process.

Network: the reference loads the pretrained ``yolov7-e6e.pt`` through the un-vendored yolov7 repository [ext]; neither
exists offline.  ``weights`` names a TorchScript export of that model (``export.py --grid`` of the yolov7 repository: output
[B, N, 85]); a path that is given must load.  Only with ``allow_random_init=True`` a random-init detector of the same I/O
contract and the same head geometry (strides 8/16/32/64, 3 anchors each, 85 channels -> 17 850 candidates at 448x640)
stands in, so that throughput is measured on an E6E-class convolutional load whose GFLOPs are stated next to E6E's."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import det_ops
from .coco_classes import COCO_CLASSES
from .detections import ObjectDetections


def _conv(c1: int, c2: int, k: int = 1, s: int = 1) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(c1, c2, k, s, k // 2, bias=False), nn.BatchNorm2d(c2), nn.SiLU(inplace=True))


class _Elan(nn.Module):
    """ELAN-style aggregation block (two 1x1 branches, a chain of 3x3 convs, concat, 1x1 fuse)."""

    def __init__(self, c1: int, c2: int, depth: int = 6):
        super().__init__()
        h = c2 // 4
        self.a, self.b = _conv(c1, h), _conv(c1, h)
        self.chain = nn.ModuleList([_conv(h, h, 3) for _ in range(depth)])
        self.fuse = _conv(h * (2 + depth // 2), c2)

    def forward(self, x):
        outs = [self.a(x), self.b(x)]
        y = outs[-1]
        for i, m in enumerate(self.chain):
            y = m(y)
            if i % 2 == 1:
                outs.append(y)
        return self.fuse(torch.cat(outs, 1))


class _Down(nn.Module):
    """DownC: max-pool + 1x1 branch beside a strided 3x3 branch."""

    def __init__(self, c1: int, c2: int):
        super().__init__()
        self.p = nn.Sequential(nn.MaxPool2d(2, 2), _conv(c1, c2 // 2))
        self.c = nn.Sequential(_conv(c1, c1), _conv(c1, c2 // 2, 3, 2))

    def forward(self, x):
        return torch.cat((self.c(x), self.p(x)), 1)


class YoloV7E6EClassNet(nn.Module):
    """E6E-class detector: ReOrg stem, five Down+ELAN stages (80..1280 channels), SPP neck, top-down/bottom-up fusion and an
    anchor head on four levels (strides 8, 16, 32, 64).  Output [B, sum(3*h*w), 5 + nc] in input pixels, like the inference
    output of yolov7's IDetect."""

    ANCHORS = [[19, 27, 44, 40, 38, 94], [96, 68, 86, 152, 180, 137], [140, 301, 303, 264, 238, 542],
               [436, 615, 739, 380, 925, 792]]
    STRIDES = [8, 16, 32, 64]

    def __init__(self, nc: int = 80, width: int = 80):
        super().__init__()
        w = width
        self.nc = nc
        self.stem = _conv(12, w, 3)
        chans = [w * 2, w * 4, w * 8, w * 12, w * 16]
        self.down = nn.ModuleList([_Down(c1, c2) for c1, c2 in zip([w] + chans[:-1], chans)])
        self.elan = nn.ModuleList([_Elan(c, c) for c in chans])
        self.spp = nn.Sequential(_conv(chans[4], chans[4] // 2), nn.MaxPool2d(5, 1, 2), _conv(chans[4] // 2, chans[4] // 2))
        head_c = [chans[1] // 2, chans[2] // 2, chans[3] // 2, chans[4] // 2]  # P3..P6
        self.lat = nn.ModuleList([_conv(chans[i + 1], head_c[i]) for i in range(3)])
        self.td = nn.ModuleList([_Elan(head_c[i] + head_c[i + 1], head_c[i], 4) for i in range(3)])
        self.bu_down = nn.ModuleList([_conv(head_c[i], head_c[i + 1], 3, 2) for i in range(3)])
        self.bu = nn.ModuleList([_Elan(2 * head_c[i + 1], head_c[i + 1], 4) for i in range(3)])
        self.detect = nn.ModuleList([nn.Conv2d(c, 3 * (5 + nc), 1) for c in head_c])
        self.register_buffer("anchors", torch.tensor(self.ANCHORS, dtype=torch.float32).view(4, 3, 2))
        # yolov7's Detect._initialize_biases [ext]: objectness prior of ~8 objects per 640-px image, class prior 0.6/nc --
        # without it a random-init head passes half of the 17 850 candidates through conf > 0.25, which no trained detector
        # does and which would turn the NMS into the dominant cost of the "full step" side figure
        import math

        with torch.no_grad():
            for m, stride in zip(self.detect, self.STRIDES):
                b = m.bias.view(3, -1)
                b[:, 4] = math.log(8 / (640 / stride) ** 2)
                b[:, 5:] = math.log(0.6 / (nc - 0.99))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        b = x.shape[0]
        x = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)  # ReOrg
        x = self.stem(x)
        feats = []
        for d, e in zip(self.down, self.elan):
            x = e(d(x))
            feats.append(x)
        p = [None, None, None, self.spp(feats[4])]
        for i in (2, 1, 0):  # top-down
            up = nn.functional.interpolate(p[i + 1], size=feats[i + 1].shape[-2:], mode="nearest")
            p[i] = self.td[i](torch.cat((self.lat[i](feats[i + 1]), up), 1))
        for i in range(3):   # bottom-up
            p[i + 1] = self.bu[i](torch.cat((self.bu_down[i](p[i]), p[i + 1]), 1))
        z = []
        for i, f in enumerate(p):
            y = self.detect[i](f)
            _, _, ny, nx = y.shape
            y = y.view(b, 3, 5 + self.nc, ny, nx).permute(0, 1, 3, 4, 2).sigmoid()
            gy, gx = torch.meshgrid(torch.arange(ny, device=y.device), torch.arange(nx, device=y.device), indexing="ij")
            grid = torch.stack((gx, gy), -1).view(1, 1, ny, nx, 2).to(y.dtype)
            xy = (y[..., 0:2] * 2.0 - 0.5 + grid) * self.STRIDES[i]
            wh = (y[..., 2:4] * 2) ** 2 * self.anchors[i].view(1, 3, 1, 1, 2).to(y.dtype)
            z.append(torch.cat((xy, wh, y[..., 4:]), -1).view(b, -1, 5 + self.nc))
        return torch.cat(z, 1)


E6E_PUBLISHED = {"params_M": 151.7, "gflops_at_1280x1280": 843.2}  # yolov7 README, YOLOv7-E6E row [ext]


def conv_gflops(model: nn.Module, example: torch.Tensor) -> float:
    """FLOPs (2 x MACs) of the convolutions of one forward of ``model`` on ``example``, in GFLOP (hooks; eager modules
    only: a TorchScript module returns 0)."""
    total = [0.0]
    hooks = []

    def hook(m, inp, out):
        k = m.kernel_size[0] * m.kernel_size[1] * (m.in_channels // m.groups)
        total[0] += 2.0 * out.numel() * k

    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(hook))
    with torch.inference_mode():
        model(example)
    for h in hooks:
        h.remove()
    return total[0] / 1e9


class YOLOv7:
    """yolov7.py:28-110 (+ ``predict_batch`` for the batched harness).

    ``weights``: the reference hands ``yolov7-e6e.pt`` to the yolov7 repository's ``attempt_load`` (yolov7.py:35), which
    unpickles that repository's own model classes -- impossible without the repository.  What this class loads instead is
    a TorchScript export of the same network whose output is the inference tensor [B, N, 85] (``python export.py
    --weights yolov7-e6e.pt --grid --img-size 448 640`` in the yolov7 repository writes ``yolov7-e6e.torchscript.pt``);
    ``YOLOV7_TORCHSCRIPT`` names it when the argument is omitted.  A path that is given must exist and must be
    TorchScript: there is no silent fallback.  ``allow_random_init=True`` builds the E6E-class stand-in network with
    random weights -- for benchmarks and pipeline tests only; ``description`` states its convolution GFLOPs next to the
    published YOLOv7-E6E figure so that "full step" throughput numbers can be read."""

    def __init__(self, weights: Optional[str] = None, image_size: int = 640, half_precision: bool = True, device=None,
                 width: int = 80, allow_random_init: bool = False) -> None:
        from ..mapping.base_map import require_gpu

        self.device = require_gpu(device)
        self.half_precision = half_precision
        self.image_size = image_size
        self.in_hw = (int(self.image_size * 0.7), self.image_size)  # (448, 640) (yolov7.py:73)
        weights = weights or os.environ.get("YOLOV7_TORCHSCRIPT")
        e6e_here = E6E_PUBLISHED["gflops_at_1280x1280"] * (self.in_hw[0] * self.in_hw[1]) / (1280.0 * 1280.0)
        if weights:
            if not os.path.isfile(weights):
                raise FileNotFoundError(f"YOLOv7 weights {weights!r} not found")
            try:
                self.model = torch.jit.load(weights, map_location=self.device).eval()
            except Exception as exc:  # noqa: BLE001 -- a pickled yolov7 checkpoint, not TorchScript
                raise ValueError(
                    f"{weights!r} is not a TorchScript module ({type(exc).__name__}).  yolov7's .pt checkpoints pickle the "
                    "yolov7 repository's own classes and can only be opened with that repository; export it once with "
                    "`python export.py --weights yolov7-e6e.pt --grid --img-size 448 640` and pass the resulting "
                    "*.torchscript.pt (or set YOLOV7_TORCHSCRIPT)") from exc
            self.weights = f"torchscript:{weights}"
            self.description = self.weights
        elif allow_random_init:
            with torch.device(self.device):
                self.model = YoloV7E6EClassNet(width=width)
            self.model.eval()
            self.weights = "random-init (E6E-class stand-in)"
            det_ops.fold_batchnorm_(self.model)   # as yolov7's fuse() does before tracing / deployment
            g = conv_gflops(self.model, torch.zeros((1, 3) + self.in_hw, device=self.device))
            self.description = (f"random-init E6E-class stand-in, width {width}: {g:.1f} conv GFLOPs per {self.in_hw[1]}x"
                                f"{self.in_hw[0]} frame (published YOLOv7-E6E: {E6E_PUBLISHED['gflops_at_1280x1280']} GFLOPs at "
                                f"1280x1280 = {e6e_here:.1f} at this input size, {E6E_PUBLISHED['params_M']} M parameters); "
                                f"stand-in / E6E = {g / e6e_here:.2f}")
            self.stand_in_gflops, self.e6e_gflops = g, e6e_here
        else:
            raise ValueError("YOLOv7 needs `weights` (a TorchScript export of yolov7-e6e, see the class docstring) or "
                             "YOLOV7_TORCHSCRIPT; pass allow_random_init=True for the random-init stand-in (benchmarks only)")
        if self.half_precision:
            self.model.half()

    @torch.inference_mode()
    def predict_batch(self, images_u8: torch.Tensor, conf_thres: float = 0.25, iou_thres: float = 0.45,
                      classes: Optional[Sequence[int]] = None, agnostic_nms: bool = False) -> List[ObjectDetections]:
        """images_u8 [B,H,W,3] u8 RGB on device -> one ObjectDetections per image (boxes normalised xyxy)."""
        B, H, W, _ = images_u8.shape
        img = det_ops.resize_area(images_u8, self.in_hw[0], self.in_hw[1],
                                  torch.float16 if self.half_precision else torch.float32)
        pred = self.model(img)
        pred = pred[0] if isinstance(pred, (tuple, list)) else pred
        dets = det_ops.non_max_suppression(pred.float(), conf_thres, iou_thres, classes=classes, agnostic=agnostic_nms)
        out = []
        for b, p in enumerate(dets):
            p = p.clone()
            if p.shape[0]:
                p[:, :4] = det_ops.scale_coords(self.in_hw, p[:, :4], (H, W, 3)).round()
                p[:, 0] /= W
                p[:, 1] /= H
                p[:, 2] /= W
                p[:, 3] /= H
            p = p.cpu()
            phrases = [COCO_CLASSES[int(i)] for i in p[:, 5]]
            out.append(ObjectDetections(p[:, :4], p[:, 4], phrases, image_source=None, fmt="xyxy"))
        return out

    def predict(self, image: np.ndarray, conf_thres: float = 0.25, iou_thres: float = 0.45,
                classes: Optional[List[str]] = None, agnostic_nms: bool = False) -> ObjectDetections:
        img = torch.from_numpy(np.ascontiguousarray(image)).to(self.device)[None]
        det = self.predict_batch(img, conf_thres, iou_thres, classes, agnostic_nms)[0]
        det.image_source = image
        return det


class YOLOv7Client:
    """yolov7.py:113-121: ``predict(image_numpy) -> ObjectDetections``; ``port`` is accepted and ignored (in-process).
    ``emulate_jpeg=True`` reproduces the reference's quality-90 JPEG transport (server_wrapper.py:57-68) for A/B checks."""

    _shared: Dict[str, YOLOv7] = {}

    def __init__(self, port: int = 12184, device=None, emulate_jpeg: bool = False, **model_kwargs) -> None:
        key = str(device)
        if key not in YOLOv7Client._shared:
            YOLOv7Client._shared[key] = YOLOv7(device=device, **model_kwargs)
        self._model = YOLOv7Client._shared[key]
        self._emulate_jpeg = emulate_jpeg
        self.url = f"inprocess://yolov7 (port {port} ignored)"

    def predict(self, image_numpy: np.ndarray) -> ObjectDetections:
        seen = image_numpy
        if self._emulate_jpeg:
            from .transport import jpeg_roundtrip

            seen = jpeg_roundtrip(image_numpy)
        # the reference round-trips through JSON (yolov7.py:117-119): float32 tensors rebuilt from Python lists; the
        # detections are attached to the caller's frame, not to the transported copy
        return ObjectDetections.from_json(self._model.predict(seen).to_json(), image_source=image_numpy)

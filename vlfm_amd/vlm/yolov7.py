"""vlfm.vlm.yolov7, MI355X in-process (reference: /root/reference/vlfm/vlm/yolov7.py:28-121).

``YOLOv7.predict(image)`` keeps the reference's pipeline -- cv2.resize to (640, 448) INTER_AREA, letterbox (a no-op at that
size), CHW, fp16, /255, model, non_max_suppression(0.25, 0.45), scale_coords + round, normalise by the ORIGINAL width and
height -- with the preprocessing and the NMS as HIP kernels (csrc/detect_ops.hip) and everything batched over the resident
environments (``predict_batch``).  ``YOLOv7Client`` keeps the client signature; the model lives in this process.

Network: the YOLOv7-E6E graph itself (``yolov7_e6e.YoloV7E6E``: the 262 modules of cfg/deploy/yolov7-e6e.yaml with the
checkpoint's own indices, 151.7 M parameters, 843 GFLOPs at 1280 x 1280).  ``weights`` is what the reference hands to
``attempt_load`` (yolov7.py:35): ``yolov7-e6e.pt`` -- opened WITHOUT the yolov7 repository through a restricted unpickler
(yolov7_e6e.read_yolov7_checkpoint) and loaded strictly (every tensor of the file consumed, every tensor of the graph fed) --
or a plain state-dict file of that model, or a TorchScript export (``export.py --grid``: output [B, N, 85]).  A path that is
given must load; only ``allow_random_init=True`` builds the same graph with random weights (benchmarks, pipeline tests)."""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import det_ops
from .coco_classes import COCO_CLASSES
from .detections import ObjectDetections
from .yolov7_e6e import PUBLISHED as E6E_PUBLISHED
from .yolov7_e6e import YoloV7E6E, count_parameters, gflops, load_yolov7_state_dict, read_yolov7_checkpoint


class YOLOv7:
    """yolov7.py:28-110 (+ ``predict_batch`` for the batched harness).

    ``weights`` (or ``YOLOV7_WEIGHTS``): ``yolov7-e6e.pt`` as downloaded for the reference (README "yolov7-e6e.pt" ->
    ``data/yolov7-e6e.pt``, yolov7.py:122), a ``state_dict`` file of it, or a TorchScript export.  There is no silent
    fallback: a path that does not exist, or holds neither, raises.  ``width_multiple`` < 1 (with ``allow_random_init``)
    builds a same-topology miniature for tests; the deployed network is 1.0."""

    def __init__(self, weights: Optional[str] = None, image_size: int = 640, half_precision: bool = True, device=None,
                 allow_random_init: bool = False, width_multiple: float = 1.0, hip_conv: bool = True) -> None:
        from ..mapping.base_map import require_gpu

        self.device = require_gpu(device)
        self.half_precision = half_precision
        self.image_size = image_size
        self.in_hw = (int(self.image_size * 0.7), self.image_size)  # (448, 640) (yolov7.py:73)
        weights = weights or os.environ.get("YOLOV7_WEIGHTS") or os.environ.get("YOLOV7_TORCHSCRIPT")
        if weights:
            if not os.path.isfile(weights):
                raise FileNotFoundError(f"YOLOv7 weights {weights!r} not found")
            self.model, self.weights = self._load(weights)
            self.description = self.weights
        elif allow_random_init:
            with torch.device(self.device):
                self.model = YoloV7E6E(width_multiple=width_multiple)
            self.model.init_random().eval()
            self.weights = "random-init yolov7-e6e graph" + ("" if width_multiple == 1.0 else f" (width x{width_multiple})")
        else:
            raise ValueError("YOLOv7 needs `weights` (yolov7-e6e.pt, a state dict of it, or a TorchScript export; or "
                             "YOLOV7_WEIGHTS); pass allow_random_init=True for the random-init graph (benchmarks only)")
        if isinstance(self.model, YoloV7E6E):
            n_params = count_parameters(self.model)
            self.model.fuse_()   # as attempt_load(..., fuse=True) does (yolov7.py:35): BatchNorm folded, bias + SiLU one pass
            self.gflops = gflops(self.model, *self.in_hw)
            self.description = (f"{self.weights}: {n_params / 1e6:.1f} M parameters, {self.gflops:.1f} conv GFLOPs per "
                                f"{self.in_hw[1]}x{self.in_hw[0]} frame (published YOLOv7-E6E: {E6E_PUBLISHED['params_M']} M, "
                                f"{E6E_PUBLISHED['gflops_at_1280x1280']} GFLOPs at 1280x1280 = "
                                f"{E6E_PUBLISHED['gflops_at_1280x1280'] * self.in_hw[0] * self.in_hw[1] / 1280.0 ** 2:.1f} here)")
        if self.half_precision:
            self.model.half()
        # fp16 on the GPU: NHWC execution with the 1x1 / 3x3 layers as implicit GEMMs on the matrix cores (csrc/conv_nhwc.hip)
        self.hip_convs = 0
        if isinstance(self.model, YoloV7E6E) and self.half_precision and hip_conv:
            self.hip_convs = self.model.use_hip_conv_()
            self.description += f"; {self.hip_convs} convolutions on conv_nhwc.hip"

    MAX_FRAMES_PER_FORWARD = 256

    def _load(self, path: str):
        try:
            return torch.jit.load(path, map_location=self.device).eval(), f"torchscript:{path}"
        except Exception as jit_exc:  # noqa: BLE001 -- not TorchScript: a yolov7 checkpoint or a state dict
            try:
                sd = read_yolov7_checkpoint(path)
                with torch.device(self.device):
                    model = YoloV7E6E()
                form = load_yolov7_state_dict(model, sd)
            except Exception as exc:  # noqa: BLE001
                raise ValueError(
                    f"{path!r} is neither a TorchScript module ({type(jit_exc).__name__}) nor a yolov7-e6e checkpoint / "
                    f"state dict ({type(exc).__name__}: {exc})") from exc
            return model.eval(), f"yolov7-e6e checkpoint ({form}):{path}"

    @torch.inference_mode()
    def predict_batch(self, images_u8: torch.Tensor, conf_thres: float = 0.25, iou_thres: float = 0.45,
                      classes: Optional[Sequence[int]] = None, agnostic_nms: bool = False,
                      pred_hook=None) -> List[ObjectDetections]:
        """images_u8 [B,H,W,3] u8 RGB on device -> one ObjectDetections per image (boxes normalised xyxy).
        ``pred_hook(pred, in_hw)``: called with the network's raw prediction [B, N, 85] (xywh in input pixels, objectness, class
        scores) before the post-processing; the benchmark harness uses it to let a SCRIPTED head speak through the real
        non_max_suppression / scale_coords path when the weights are random (vlfm_amd/harness.py)."""
        B, H, W, _ = images_u8.shape
        img = det_ops.resize_area(images_u8, self.in_hw[0], self.in_hw[1],
                                  torch.float16 if self.half_precision else torch.float32)
        if self.hip_convs:
            img = img.contiguous(memory_format=torch.channels_last)
        # csrc/conv_nhwc.hip addresses with 32-bit element offsets: the widest activation (80 channels at half resolution) of 256
        # frames is 1.47 G elements, so larger batches go through the network in slices
        preds = []
        for i in range(0, B, self.MAX_FRAMES_PER_FORWARD):
            pred = self.model(img[i:i + self.MAX_FRAMES_PER_FORWARD])
            preds.append(pred[0] if isinstance(pred, (tuple, list)) else pred)
        pred = preds[0] if len(preds) == 1 else torch.cat(preds, 0)
        if pred_hook is not None:
            pred = pred_hook(pred, self.in_hw)
        dets = det_ops.non_max_suppression(pred.float(), conf_thres, iou_thres, classes=classes, agnostic=agnostic_nms)
        # rescale / round / normalise all detections of the batch at once, one transfer to the host (yolov7.py:99-110 per image)
        sizes = [int(p.shape[0]) for p in dets]
        allp = torch.cat(dets, 0).clone() if sum(sizes) else torch.zeros((0, 6))
        if allp.shape[0]:
            allp[:, :4] = det_ops.scale_coords(self.in_hw, allp[:, :4], (H, W, 3)).round()
            allp[:, 0] /= W
            allp[:, 1] /= H
            allp[:, 2] /= W
            allp[:, 3] /= H
            allp = allp.cpu()
        out, start = [], 0
        for n in sizes:
            p = allp[start:start + n]
            start += n
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

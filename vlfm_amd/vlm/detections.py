"""vlfm.vlm.detections.ObjectDetections, drop-in (reference: /root/reference/vlfm/vlm/detections.py:15-126).

Container for detector outputs with the reference's constructor, attributes and filters.  ``box_convert`` restates
torchvision.ops.box_convert [ext] (torchvision is not installed here); drawing (``annotate``) is visualisation and out of
scope -- ``annotated_frame`` returns the untouched ``image_source``."""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch


def box_convert(boxes: torch.Tensor, in_fmt: str, out_fmt: str) -> torch.Tensor:
    """torchvision.ops.box_convert for the formats the path uses (detections.py:30-33)."""
    if in_fmt == out_fmt:
        return boxes.clone()
    if in_fmt == "cxcywh" and out_fmt == "xyxy":
        cx, cy, w, h = boxes.unbind(-1)
        return torch.stack((cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h), dim=-1)
    if in_fmt == "xywh" and out_fmt == "xyxy":
        x, y, w, h = boxes.unbind(-1)
        return torch.stack((x, y, x + w, y + h), dim=-1)
    if in_fmt == "xyxy" and out_fmt == "cxcywh":
        x1, y1, x2, y2 = boxes.unbind(-1)
        return torch.stack(((x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1), dim=-1)
    raise ValueError(f"unsupported conversion {in_fmt} -> {out_fmt}")


class ObjectDetections:
    """detections.py:15-126: boxes are normalised xyxy after construction."""

    def __init__(self, boxes: torch.Tensor, logits: torch.Tensor, phrases: List[str],
                 image_source: Optional[np.ndarray], fmt: str = "cxcywh"):
        self.image_source = image_source
        self.boxes = box_convert(boxes=boxes, in_fmt=fmt, out_fmt="xyxy") if fmt != "xyxy" else boxes
        self.logits = logits
        self.phrases = phrases
        self._annotated_frame: Optional[np.ndarray] = None

    @property
    def annotated_frame(self) -> Optional[np.ndarray]:
        return self.image_source

    @property
    def num_detections(self) -> int:
        return len(self.phrases)

    def __repr__(self) -> str:
        dets = [f"{phrase} ({logit:.2f}): {box.tolist()}"
                for box, logit, phrase in zip(self.boxes, self.logits, self.phrases)]
        return "\n".join(dets) if dets else "No detections"

    def filter_by_conf(self, conf_thresh: float) -> None:
        self._filter(torch.ge(self.logits, conf_thresh))  # >= (detections.py:70)

    def filter_by_class(self, classes: List[str]) -> None:
        self._filter(torch.tensor([p in classes for p in self.phrases], dtype=torch.bool))

    def _filter(self, keep: torch.Tensor) -> None:
        if keep.all():  # also the empty case
            return
        keep = keep.to(self.boxes.device) if torch.is_tensor(self.boxes) else keep
        self.boxes = self.boxes[keep]
        self.logits = self.logits[keep]
        self.phrases = [p for i, p in enumerate(self.phrases) if keep[i]]
        self._annotated_frame = None

    def to_json(self) -> dict:
        return {"boxes": self.boxes.tolist(), "logits": self.logits.tolist(), "phrases": self.phrases}

    @classmethod
    def from_json(cls, json_dict: dict, image_source: Optional[np.ndarray] = None) -> "ObjectDetections":
        return cls(image_source=image_source, boxes=torch.tensor(json_dict["boxes"]),
                   logits=torch.tensor(json_dict["logits"]), phrases=json_dict["phrases"], fmt="xyxy")

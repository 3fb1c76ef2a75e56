"""oracle/ref_detect.py -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the third-party arithmetic behind the detector wrappers (none of it is in /root/reference):
cv2.resize(INTER_AREA) [ext OpenCV 4.5.5 resizeArea_], torchvision.ops.nms [ext], yolov7's pre/post-processing as the
reference calls it (/root/reference/vlfm/vlm/yolov7.py:70-104).  PARITY UNPINNED against the real packages."""
from __future__ import annotations

import numpy as np


def area_tab(ssize: int, dsize: int):
    """cv::computeResizeAreaTab: list of (dst, src, alpha) in order."""
    scale = ssize / dsize
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = int(np.ceil(fsx1)), int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def resize_area_u8(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_AREA) for u8 HWC, general (non-integer-scale) path:
    float32 accumulation row by row, cvRound + saturate."""
    H, W, C = img.shape
    xt, yt = area_tab(W, out_w), area_tab(H, out_h)
    out = np.zeros((out_h, out_w, C), np.uint8)
    src = img.astype(np.float32)
    # horizontal sums per source row
    rows = np.zeros((H, out_w, C), np.float32)
    for dx, sx, a in xt:  # in table order: left to right per dx
        rows[:, dx] = rows[:, dx] + src[:, sx] * a
    acc = np.zeros((out_h, out_w, C), np.float32)
    first = np.ones(out_h, bool)
    for dy, sy, b in yt:
        if first[dy]:
            acc[dy] = b * rows[sy]
            first[dy] = False
        else:
            acc[dy] = acc[dy] + b * rows[sy]
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def nms(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    """torchvision.ops.nms: greedy, by descending score, suppress IoU > thr (float32 arithmetic)."""
    boxes = boxes.astype(np.float32)
    order = np.argsort(-scores, kind="stable")
    keep = []
    dead = np.zeros(len(boxes), bool)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for oi, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        rest = order[oi + 1:]
        w = np.maximum(np.minimum(boxes[i, 2], boxes[rest, 2]) - np.maximum(boxes[i, 0], boxes[rest, 0]), np.float32(0))
        h = np.maximum(np.minimum(boxes[i, 3], boxes[rest, 3]) - np.maximum(boxes[i, 1], boxes[rest, 1]), np.float32(0))
        inter = w * h
        iou = inter / (area[i] + area[rest] - inter)
        dead[rest[iou > np.float32(thr)]] = True
    return np.array(keep, np.int64)

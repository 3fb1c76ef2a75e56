"""MobileSAM (vlfm/vlm/sam.py:40-57) on a batch of frames: wall time per call; run under rocprofv3 for the kernel table.
Usage: python tools/sam_probe.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from vlfm_amd.vlm.sam import MobileSAM  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
sam = MobileSAM(device=dev, allow_random_init=True)
rgb = torch.randint(0, 255, (n, 480, 640, 3), dtype=torch.uint8, device=dev)
box = torch.tensor([[[0.3 * 640, 0.3 * 480, 0.7 * 640, 0.8 * 480]]] * n)
for _ in range(3):
    sam.segment_bboxes(rgb, box)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    sam.segment_bboxes(rgb, box)
torch.cuda.synchronize()
print(f"MobileSAM segment_bboxes, {n} frames: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")

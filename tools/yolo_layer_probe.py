"""Per-layer-shape time of the yolov7-e6e forward on csrc/conv_nhwc.hip: HIP events around every det_ops.conv_nhwc call of a few
forwards, aggregated by (Cin, Cout, k, stride, Hout, Wout), plus the time outside the convolutions (concatenation, pooling ...).
Usage: python tools/yolo_layer_probe.py [batch]"""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from vlfm_amd.vlm import det_ops  # noqa: E402
from vlfm_amd.vlm.yolov7 import YOLOv7  # noqa: E402


def main(batch, reps=5):
    dev = torch.device("cuda:0")
    det = YOLOv7(device=dev, allow_random_init=True)
    x = torch.rand(batch, 3, 448, 640, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
    with torch.inference_mode():
        for _ in range(2):
            det.model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            det.model(x)
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) / reps
        spans = []
        inner = det_ops.conv_nhwc

        def timed(xx, w_rows, bias, ksize, stride=1, act="silu", out=None):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            y = inner(xx, w_rows, bias, ksize, stride, act, out)
            b.record()
            spans.append(((xx.shape[1], w_rows.shape[0], ksize, stride, y.shape[2], y.shape[3]), a, b))
            return y

        det_ops.conv_nhwc = timed
        for _ in range(reps):
            det.model(x)
        torch.cuda.synchronize()
        det_ops.conv_nhwc = inner
    agg = collections.OrderedDict()
    for key, a, b in spans:
        rec = agg.setdefault(key, [0, 0.0])
        rec[0] += 1
        rec[1] += a.elapsed_time(b) / reps
    conv_ms = sum(v[1] for v in agg.values())
    print(f"batch {batch}: forward {total * 1e3:.2f} ms, convolutions {conv_ms:.2f} ms (event spans incl. launch gaps), "
          f"{det.gflops * batch / (total * 1e3):.0f} TFLOP/s over the forward")
    print(f"{'Cin':>5s} {'Cout':>5s} k s {'HxW':>9s} {'calls':>5s} {'ms':>8s} {'share':>6s} {'TFLOP/s':>8s}")
    for (cin, cout, k, s, h, w), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        calls = n // reps
        flop = 2.0 * batch * h * w * cout * cin * k * k * calls
        print(f"{cin:5d} {cout:5d} {k} {s} {h:4d}x{w:<4d} {calls:5d} {ms:8.3f} {ms / conv_ms * 100:5.1f}% {flop / ms / 1e9:8.0f}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 32)

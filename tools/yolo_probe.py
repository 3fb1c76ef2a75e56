"""YOLOv7-E6E forward time at the reference's input size (448x640, f16) for a batch of frames: NCHW (the path YOLOv7 runs:
folded BatchNorm, bias + SiLU by vlfm_bias_act_nchw in place) against channels_last (framework ops for bias + SiLU), with and
without MIOpen's find mode.  Usage: python tools/yolo_probe.py [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from vlfm_amd.vlm.yolov7 import YOLOv7  # noqa: E402


def timed(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main(batch):
    dev = torch.device("cuda:0")
    hip = YOLOv7(device=dev, allow_random_init=True)
    print(hip.description, flush=True)
    x = torch.rand(batch, 3, 448, 640, device=dev, dtype=torch.float16)
    xcl = x.contiguous(memory_format=torch.channels_last)
    det = YOLOv7(device=dev, allow_random_init=True, hip_conv=False)
    with torch.inference_mode():
        torch.backends.cudnn.benchmark = "hip-only" not in sys.argv   # the layers left on the framework: MIOpen's find mode
        t = timed(lambda: hip.model(xcl))
        print(f"NHWC, conv_nhwc.hip ({hip.hip_convs} layers)     : {t:8.2f} ms / {batch} frames = "
              f"{hip.gflops * batch / t:.0f} TFLOP/s", flush=True)
        if "hip-only" in sys.argv:
            return
        a, b = hip.model(xcl)[0].float(), det.model(x)[0].float()
        print(f"  max |difference| of the raw predictions against the NCHW framework path: {float((a - b).abs().max()):.4g} "
              f"(values up to {float(b.abs().max()):.4g})", flush=True)
        torch.backends.cudnn.benchmark = False
        print(f"NCHW, BiasAct kernel in place        : {timed(lambda: det.model(x)):8.2f} ms / {batch} frames", flush=True)
        if len(sys.argv) > 2 and sys.argv[2] == "nchw-only":
            return
        torch.backends.cudnn.benchmark = True
        print(f"NCHW, + MIOpen find mode             : {timed(lambda: det.model(x)):8.2f} ms", flush=True)
        torch.backends.cudnn.benchmark = False
        for m in det.model.modules():      # framework bias + SiLU (works for any memory format)
            if hasattr(m, "inplace") and type(m).__name__ == "BiasAct":
                m.inplace = False
        print(f"NCHW, framework bias + SiLU          : {timed(lambda: det.model(x)):8.2f} ms", flush=True)
        det.model.to(memory_format=torch.channels_last)
        xc = x.contiguous(memory_format=torch.channels_last)
        print(f"channels_last, framework bias + SiLU : {timed(lambda: det.model(xc)):8.2f} ms", flush=True)
        torch.backends.cudnn.benchmark = True
        print(f"channels_last, + MIOpen find mode    : {timed(lambda: det.model(xc)):8.2f} ms", flush=True)
    g = det.gflops * batch
    print(f"({g / 1e3:.2f} TFLOP per batch)")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 32)

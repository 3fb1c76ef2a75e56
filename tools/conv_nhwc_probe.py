"""vlfm_conv_nhwc_f16 against the framework's convolution (MIOpen, channels_last, + the separate bias + SiLU pass) on the heaviest
layer shapes of the yolov7-e6e graph at 448x640.  Usage: python tools/conv_nhwc_probe.py [batch] [sweep]   (sweep: every tile shape of the kernel on every probe shape)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from vlfm_amd.vlm import det_ops  # noqa: E402

SHAPES = [(64, 64, 3, 1, 112, 160), (128, 128, 3, 1, 56, 80), (256, 256, 3, 1, 28, 40), (384, 384, 3, 1, 14, 20),
          (128, 128, 3, 1, 28, 40), (512, 512, 3, 1, 7, 10), (192, 192, 3, 1, 14, 20), (320, 640, 3, 1, 28, 40),
          (640, 256, 1, 1, 28, 40), (320, 160, 1, 1, 112, 160), (1280, 640, 1, 1, 28, 40), (320, 128, 1, 1, 56, 80),
          (320, 320, 3, 2, 56, 80)]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


NCFG = 7


def sweep(batch):
    """Every tile shape (VLFM_CONV_CFG 0..6) and the library's own pick on every probe shape."""
    dev = torch.device("cuda:0")
    shapes = SHAPES + [(256, 256, 3, 1, 7, 10), (1280, 512, 1, 1, 7, 10), (64, 64, 3, 1, 56, 80), (960, 384, 1, 1, 14, 20),
                       (80, 80, 1, 1, 224, 320), (160, 64, 1, 1, 112, 160)]
    print(f"batch {batch}: us per call under each tile shape (256x256 256x128 256x64 128x128 128x64 64x128 64x64 | auto)")
    for cin, cout, k, s, H, W in shapes:
        x = torch.randn(batch, cin, H * s, W * s, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(cout, cin, k, k, device=dev, dtype=torch.float16) / (cin * k * k) ** 0.5
        rows, bias = det_ops.pack_conv_weight(w, torch.randn(cout, device=dev))
        flop = 2.0 * batch * H * W * cout * cin * k * k
        ts = []
        for cfg in list(range(NCFG)) + [None]:
            if cfg is None:
                os.environ.pop("VLFM_CONV_CFG", None)
            else:
                os.environ["VLFM_CONV_CFG"] = str(cfg)
            ts.append(timed(lambda: det_ops.conv_nhwc(x, rows, bias, k, s, "silu"), n=6))
        best = min(range(NCFG), key=lambda i: ts[i])
        print(f"{cin:5d}->{cout:5d} k{k} s{s} {H:4d}x{W:<4d}: " + " ".join(f"{t * 1e6:7.1f}" for t in ts[:NCFG])
              + f" | {ts[NCFG] * 1e6:7.1f}  best cfg {best} = {flop / ts[best] / 1e12:5.0f} TFLOP/s, auto {flop / ts[NCFG] / 1e12:5.0f}",
              flush=True)


def main(batch):
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    print(f"batch {batch}; VLFM_CONV_BN={os.environ.get('VLFM_CONV_BN', 'auto')}")
    for cin, cout, k, s, H, W in SHAPES:
        Hin, Win = H * s, W * s
        x = torch.randn(batch, cin, Hin, Win, device=dev, dtype=torch.float16).contiguous(memory_format=torch.channels_last)
        conv = torch.nn.Conv2d(cin, cout, k, s, k // 2, bias=False).to(dev).half().to(memory_format=torch.channels_last)
        b = torch.randn(cout, device=dev, dtype=torch.float16)
        rows, bias = det_ops.pack_conv_weight(conv.weight, b)
        flop = 2.0 * batch * H * W * cout * cin * k * k
        with torch.inference_mode():
            t_hip = timed(lambda: det_ops.conv_nhwc(x, rows, bias, k, s, "silu"))
            t_lib = timed(lambda: torch.nn.functional.silu(conv(x) + b.view(1, -1, 1, 1)))
            t_lib_conv = timed(lambda: conv(x))
        print(f"{cin:5d}->{cout:5d} k{k} s{s} {H:4d}x{W:<4d}: hip {t_hip * 1e6:8.1f} us = {flop / t_hip / 1e12:6.1f} TFLOP/s | "
              f"MIOpen conv {t_lib_conv * 1e6:8.1f} us = {flop / t_lib_conv / 1e12:6.1f} TFLOP/s, + bias + SiLU {t_lib * 1e6:8.1f} us",
              flush=True)


if __name__ == "__main__":
    if "sweep" in sys.argv:
        with torch.inference_mode():
            sweep(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64)
    else:
        main(int(sys.argv[1]) if len(sys.argv) > 1 else 32)

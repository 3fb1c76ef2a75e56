"""BLIP-2 preprocessing: the one-launch resampler against the two-launch form (bit-exact? how fast?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.set_num_threads(1)
from vlfm_amd.vlm import ops

dev = torch.device("cuda:0")


def timed(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for (n, H, W, out, patch, dt) in [(256, 480, 640, 224, 14, torch.float16), (16, 720, 1280, 224, 14, torch.float16),
                                  (256, 480, 640, 224, 0, torch.float32), (5, 120, 160, 56, 0, torch.bfloat16),
                                  (3, 96, 128, 224, 14, torch.float16)]:
    img = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device=dev)
    os.environ["VLFM_PREPROCESS_TWO_PASS"] = "1"
    want = ops.preprocess_rgb(img, out, dt, patch_size=patch)
    t_old = timed(lambda: ops.preprocess_rgb(img, out, dt, patch_size=patch))
    del os.environ["VLFM_PREPROCESS_TWO_PASS"]
    got = ops.preprocess_rgb(img, out, dt, patch_size=patch)
    t_new = timed(lambda: ops.preprocess_rgb(img, out, dt, patch_size=patch))
    same = torch.equal(got, want)
    mb = (img.numel() + got.numel() * got.element_size()) / 1e6
    print(f"{n}x{H}x{W}->{out} patch {patch} {dt}: equal={same}  two-pass {t_old:.1f} us  fused {t_new:.1f} us "
          f"({mb / t_new:.2f} TB/s over {mb:.0f} MB)", flush=True)
    assert same
